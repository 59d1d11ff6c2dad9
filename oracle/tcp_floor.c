/* TEST / BASELINE INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The floor of what the reference's TCP platform (GRPC_PLATFORM_TYPE=TCP, tcp_posix.cc) can
 * reach on this host: the raw sendmsg()/recvmsg() loop of tcp_flush / tcp_do_read
 * (src/core/lib/iomgr/tcp_posix.cc: iovec of the outgoing slices, up to MAX_WRITE_IOVEC = 1000
 * per sendmsg; recvmsg into the incoming slices) over a loop-back TCP connection, with no
 * HTTP/2 transport, no serialization, no completion queue on top.  gRPC-over-TCP is slower
 * than this on the same cores; bench.py quotes both this floor and a real grpcio loop-back run.
 *
 *   tcp_floor stream   <n_msgs> <payload_bytes>   client-streaming shape: every message is one
 *                                                 sendmsg of [9-byte header][<= 16384 payload]...
 *   tcp_floor pingpong <iters> <bytes>            unary shape: write <bytes>, read <bytes> back
 * Two threads (writer + reader / client + server) = 2 cores.  Prints one JSON line.
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void die(const char* what) {
  perror(what);
  exit(2);
}

static int listen_loopback(uint16_t* port) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) die("socket");
  struct sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  if (bind(fd, (struct sockaddr*)&a, sizeof a) != 0) die("bind");
  socklen_t len = sizeof a;
  if (getsockname(fd, (struct sockaddr*)&a, &len) != 0) die("getsockname");
  *port = ntohs(a.sin_port);
  if (listen(fd, 1) != 0) die("listen");
  return fd;
}

static int connect_loopback(uint16_t port) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) die("socket");
  struct sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  a.sin_port = htons(port);
  if (connect(fd, (struct sockaddr*)&a, sizeof a) != 0) die("connect");
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);  /* tcp_posix sets TCP_NODELAY too */
  return fd;
}

struct rx_arg {
  int lfd;
  uint64_t total;    /* stream: bytes to receive */
  uint64_t iters;    /* pingpong: messages to echo */
  size_t bytes;
};

static void* stream_reader(void* p) {
  struct rx_arg* a = (struct rx_arg*)p;
  int fd = accept(a->lfd, NULL, NULL);
  if (fd < 0) die("accept");
  /* tcp_posix reads into slices sized by its estimate; a 1 MiB target is its steady state here */
  size_t cap = 1 << 20;
  uint8_t* buf = (uint8_t*)malloc(cap);
  uint64_t got = 0;
  while (got < a->total) {
    struct iovec iov = {buf, cap};
    struct msghdr m;
    memset(&m, 0, sizeof m);
    m.msg_iov = &iov;
    m.msg_iovlen = 1;
    ssize_t n = recvmsg(fd, &m, 0);
    if (n < 0 && errno == EINTR) continue;
    if (n <= 0) die("recvmsg");
    got += (uint64_t)n;
  }
  uint8_t ack = 1;
  if (write(fd, &ack, 1) != 1) die("ack");
  free(buf);
  close(fd);
  return NULL;
}

static void* echo_server(void* p) {
  struct rx_arg* a = (struct rx_arg*)p;
  int fd = accept(a->lfd, NULL, NULL);
  if (fd < 0) die("accept");
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  uint8_t* buf = (uint8_t*)malloc(a->bytes);
  for (uint64_t i = 0; i < a->iters; i++) {
    size_t got = 0;
    while (got < a->bytes) {
      ssize_t n = recv(fd, buf + got, a->bytes - got, 0);
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) die("recv");
      got += (size_t)n;
    }
    size_t put = 0;
    while (put < a->bytes) {
      ssize_t n = send(fd, buf + put, a->bytes - put, MSG_NOSIGNAL);
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) die("send");
      put += (size_t)n;
    }
  }
  free(buf);
  close(fd);
  return NULL;
}

static int cmp_d(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : x > y;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s stream <n_msgs> <payload> | pingpong <iters> <bytes>\n", argv[0]);
    return 1;
  }
  uint16_t port = 0;
  int lfd = listen_loopback(&port);
  pthread_t th;
  if (strcmp(argv[1], "stream") == 0) {
    const uint64_t n_msgs = strtoull(argv[2], NULL, 10);
    const size_t payload = strtoull(argv[3], NULL, 10);
    /* slices of one message: [9][<=16384] pairs over 5 + 1 + varint + payload bytes */
    const size_t body = payload + 5 + 1 + 4;
    const size_t nfr = (body + 16383) / 16384;
    struct iovec* iov = (struct iovec*)calloc(2 * nfr, sizeof *iov);
    uint8_t* data = (uint8_t*)malloc(body);
    uint8_t* hdrs = (uint8_t*)calloc(nfr, 9);
    for (size_t i = 0; i < body; i++) data[i] = (uint8_t)((i * 7 + 3) % 251);
    size_t per_msg = 0;
    for (size_t f = 0, off = 0; f < nfr; f++) {
      const size_t n = body - off < 16384 ? body - off : 16384;
      iov[2 * f].iov_base = hdrs + 9 * f;
      iov[2 * f].iov_len = 9;
      iov[2 * f + 1].iov_base = data + off;
      iov[2 * f + 1].iov_len = n;
      off += n;
      per_msg += 9 + n;
    }
    struct rx_arg ra = {lfd, (uint64_t)per_msg * n_msgs, 0, 0};
    pthread_create(&th, NULL, stream_reader, &ra);
    int fd = connect_loopback(port);
    const double t0 = now_s();
    for (uint64_t i = 0; i < n_msgs; i++) {
      /* tcp_flush: one sendmsg per pass over the remaining iovecs, partial sends advance the cursor */
      size_t first = 0, first_off = 0, left = per_msg;
      struct iovec tmp[1000];
      while (left) {
        size_t cnt = 0;
        for (size_t k = first; k < 2 * nfr && cnt < 1000; k++, cnt++) {
          tmp[cnt] = iov[k];
          if (k == first) {
            tmp[cnt].iov_base = (uint8_t*)iov[k].iov_base + first_off;
            tmp[cnt].iov_len -= first_off;
          }
        }
        struct msghdr m;
        memset(&m, 0, sizeof m);
        m.msg_iov = tmp;
        m.msg_iovlen = cnt;
        ssize_t n = sendmsg(fd, &m, MSG_NOSIGNAL);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) die("sendmsg");
        left -= (size_t)n;
        size_t adv = (size_t)n;
        while (adv) {
          const size_t rem = iov[first].iov_len - first_off;
          if (adv >= rem) {
            adv -= rem;
            first++;
            first_off = 0;
          } else {
            first_off += adv;
            adv = 0;
          }
        }
      }
    }
    uint8_t ack;
    if (read(fd, &ack, 1) != 1) die("ack");  /* everything has been received */
    const double sec = now_s() - t0;
    pthread_join(th, NULL);
    printf("{\"mode\": \"stream\", \"msgs\": %llu, \"payload\": %zu, \"iov_per_sendmsg\": %zu, \"seconds\": %.6f, "
           "\"GiBps\": %.4f, \"threads\": 2}\n",
           (unsigned long long)n_msgs, payload, 2 * nfr, sec, (double)payload * (double)n_msgs / sec / (double)(1ull << 30));
    close(fd);
  } else {
    const uint64_t iters = strtoull(argv[2], NULL, 10);
    const size_t bytes = strtoull(argv[3], NULL, 10);
    struct rx_arg ra = {lfd, 0, iters, bytes};
    pthread_create(&th, NULL, echo_server, &ra);
    int fd = connect_loopback(port);
    uint8_t* buf = (uint8_t*)calloc(1, bytes);
    double* rtt = (double*)malloc(sizeof(double) * iters);
    for (uint64_t i = 0; i < iters; i++) {
      const double t0 = now_s();
      size_t put = 0;
      while (put < bytes) {
        ssize_t n = send(fd, buf + put, bytes - put, MSG_NOSIGNAL);
        if (n <= 0) die("send");
        put += (size_t)n;
      }
      size_t got = 0;
      while (got < bytes) {
        ssize_t n = recv(fd, buf + got, bytes - got, 0);
        if (n <= 0) die("recv");
        got += (size_t)n;
      }
      rtt[i] = now_s() - t0;
    }
    pthread_join(th, NULL);
    qsort(rtt, iters, sizeof(double), cmp_d);
    printf("{\"mode\": \"pingpong\", \"iters\": %llu, \"bytes\": %zu, \"p50_us\": %.2f, \"p95_us\": %.2f, \"p99_us\": %.2f, "
           "\"threads\": 2}\n",
           (unsigned long long)iters, bytes, 1e6 * rtt[iters / 2], 1e6 * rtt[(size_t)((double)iters * 0.95)],
           1e6 * rtt[(size_t)((double)iters * 0.99)]);
    close(fd);
  }
  close(lfd);
  return 0;
}
