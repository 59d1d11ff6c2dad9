"""MI355X-native data plane for the gRPC RDMA_BP / RDMA_BPEV ring-buffer endpoint.

Python here is test/bench plumbing over the C ABI (include/grdma_amd.h); the
product is libgrdma_amd.so (HIP kernels + C++ host layer under csrc/).
The directory is named `grpc-rdma_amd`; import it as `grpc_rdma_amd`
(see grpc_rdma_amd.py at the repository root).
"""
from ._lib import GrdmaError, init, load  # noqa: F401
from .pair import Pair, DeviceBuffer, connect_pairs, poll_pairs, pingpong  # noqa: F401
