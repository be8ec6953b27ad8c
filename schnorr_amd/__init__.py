"""MI355X batch Schnorr-verify engine for dusk-schnorr (host layer).

    from schnorr_amd import engine as E
    E.init(0)                                   # dsv_init: select GPU, build fixed-base tables
    ok = E.verify_single(u, R, PK, m)           # numpy uint8 host arrays
    E.verify_single_dev(u, R, PK, m, ok, ws)    # torch uint8 CUDA tensors (enqueue only)

All arithmetic runs in schnorr_amd/libdsv.so (HIP, gfx950; C ABI in include/dsv.h).  There is no
CPU fallback: without the library or without a GPU every call raises.
"""
__version__ = "0.1.0"
