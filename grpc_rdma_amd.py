"""Import shim: the package directory is `grpc-rdma_amd/` (not a valid Python
identifier), so this module makes it importable as `grpc_rdma_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "grpc-rdma_amd")]
__package__ = __name__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
