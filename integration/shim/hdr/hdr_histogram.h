/* Syntax-check shim for third_party HdrHistogram_c: the declarations grpcpp/stats_time.h names. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
struct hdr_histogram;
int hdr_init(int64_t lowest, int64_t highest, int significant_figures, struct hdr_histogram** result);
void hdr_close(struct hdr_histogram* h);
void hdr_reset(struct hdr_histogram* h);
int64_t hdr_add(struct hdr_histogram* h, const struct hdr_histogram* from);
bool hdr_record_value(struct hdr_histogram* h, int64_t value);
int64_t hdr_value_at_percentile(const struct hdr_histogram* h, double percentile);
int64_t hdr_max(const struct hdr_histogram* h);
int64_t hdr_min(const struct hdr_histogram* h);
double hdr_mean(const struct hdr_histogram* h);
#ifdef __cplusplus
}
#endif
