"""md_neighbor_list_amd -- MI355X-native Verlet neighbour-list builder.

The product is libnl_hip.so (HIP kernels + C ABI, include/nl_hip.h).  This package is its Python host side:
``NeighListGPU`` mirrors the reference's class surface, ``inputs`` generates the synthetic boxes, ``slab`` does
the multi-GPU domain decomposition over torch.distributed.  Importing the package does not load the library;
constructing a builder does, and fails loudly if it has not been built.
"""
from . import inputs  # noqa: F401

__all__ = ["NeighListGPU", "NLError", "inputs"]


def __getattr__(name):
    if name in ("NeighListGPU", "NLError", "device_count"):
        from . import neighlist

        return getattr(neighlist, name)
    raise AttributeError(name)
