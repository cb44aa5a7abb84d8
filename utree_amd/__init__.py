"""utree_amd -- MI355X-native SEARCH_GG path of UTree (xtree-searchGG) behind a C-ABI.

Layout:
    csrc/            HIP kernels (gfx950) + C host orchestration  -> libutree_amd.so, xtree-searchGG
    lib.py           ctypes binding of include/utree_amd.h (fails loudly when the .so is missing)
    search.py        host-side mirror of the reference's seams (XT_read32 / XT_doSearch32 / XT_getIX32)
    ctrfile.py       numpy reader/writer of the unchanged `.ctr` format (tests, synthetic DBs)
    synth.py         seeded synthetic databases and reads for bench.py

There is no CPU fallback anywhere in this package.
"""
__all__ = ["lib", "search", "ctrfile", "synth"]
