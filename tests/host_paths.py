"""Host-only entry points of libegot2x (workspace layouts, configuration validation, error paths, implementation / slice policy, the
RCCL binding's resolver) driven through ctypes WITHOUT torch and without a GPU. Run two ways:

  * imported by tests/test_cpu_host.py against the product library;
  * as a script in a subprocess with the ASAN runtime preloaded against the host-sanitized build (egot2_amd/build.py build_sanitized:
    AddressSanitizer + UBSan on the C++ orchestration, never on the GPU): `python tests/host_paths.py <lib.so>`; any report aborts.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def bind(path):
    from egot2_amd import _lib
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.egx_abi_version() == _lib.EGX_ABI_VERSION
    return lib


def exercise(lib) -> int:
    """Returns the number of calls made; raises AssertionError on a wrong answer."""
    from egot2_amd._lib import Config, DecConfig, Segment
    n = 0
    sv, sc = C.c_size_t(), C.c_size_t()

    def segs_of(dims, T=15, proj=True):
        s = (Segment * len(dims))()
        for x, k in zip(s, dims):
            x.T, x.d_in, x.proj_w = T, k, (1 if proj else 0)
        return s

    # every implementation's layout: per-clip (one workgroup, sliced, cut), tiled, wide, generic; all compute modes; odd batches
    shapes = [
        (128, 4, 2048, 1, [256] * 3, 15), (128, 4, 2048, 2, [256] * 3, 15), (128, 8, 256, 6, [8192, 8192, 2048, 256], 12),
        (128, 4, 2048, 1, [256] * 3, 60), (128, 4, 2048, 2, [256] * 2, 150), (128, 4, 384, 1, [256] * 3, 15),
        (256, 4, 2048, 3, [256] * 3, 15), (512, 8, 2048, 3, [8192, 8192, 2048, 256], 12), (768, 8, 2048, 4, [8192, 8192, 768, 2048], 32),
        (96, 4, 200, 1, [100, 60], 7),
    ]
    for d, h, dff, L, dims, T in shapes:
        for compute in (0, 1, 2):
            for impl in (0, 1, 2, 3, 4):
                for B in (1, 7, 26, 32, 129, 256, 1000):
                    for det in (0, 1):
                        cfg = Config(d, h, dff, L, len(dims), 1e-5, compute, impl, 0.1, 0.1, 0.0)
                        cfg.deterministic = det
                        segs = segs_of(dims, T)
                        rc = lib.egx_encoder_workspace(C.byref(cfg), segs, B, C.byref(sv), C.byref(sc))
                        rc2 = lib.egx_translator_workspace(C.byref(cfg), segs, B, C.byref(sv), C.byref(sc))
                        im = lib.egx_encoder_impl(C.byref(cfg), segs, B)
                        sl = lib.egx_encoder_slices(C.byref(cfg), segs, B)
                        uf = lib.egx_encoder_uses_fused(C.byref(cfg), segs, B)
                        n += 5
                        if rc == 0:
                            assert sv.value > 0 and rc2 == 0
                        else:
                            assert lib.egx_last_error()
                        assert im in (-1, 1, 2, 3, 4) and sl in (-1, 1, 2, 4, 8) and uf in (0, 1)
                        if impl == 0 and rc == 0:
                            assert im >= 1, "auto must always find an implementation"
    # error paths
    cfg = Config(128, 4, 2048, 1, 3, 1e-5, 0, 0, 0.0, 0.0, 0.0)
    segs = segs_of([256] * 3)
    for bad, frag in ((Config(132, 5, 2048, 1, 3, 1e-5, 0, 0, 0.0, 0.0, 0.0), b"n_heads"), (Config(128, 4, 2048, 1, 0, 1e-5, 0, 0, 0.0, 0.0, 0.0), b"n_segments"),
                      (Config(128, 4, 2048, 1, 99, 1e-5, 0, 0, 0.0, 0.0, 0.0), b"n_segments"), (Config(128, 4, 2048, 999, 3, 1e-5, 0, 0, 0.0, 0.0, 0.0), b"n_layers")):
        assert lib.egx_encoder_workspace(C.byref(bad), segs, 8, C.byref(sv), C.byref(sc)) != 0 and frag in lib.egx_last_error()
        n += 1
    assert lib.egx_encoder_workspace(C.byref(cfg), segs, 0, C.byref(sv), C.byref(sc)) != 0
    assert lib.egx_encoder_workspace(C.byref(cfg), segs, -3, C.byref(sv), C.byref(sc)) != 0
    assert lib.egx_encoder_workspace(None, segs, 8, C.byref(sv), C.byref(sc)) != 0
    assert lib.egx_encoder_workspace(C.byref(cfg), None, 8, C.byref(sv), C.byref(sc)) != 0
    # null-pointer validation of the compute entry points (they must refuse before touching the device)
    assert lib.egx_encoder_fwd(C.byref(cfg), segs, None, None, None, 8, None, None, None, 0, 0, None) != 0
    assert lib.egx_translator_fwd(C.byref(cfg), segs, None, None, None, None, 8, None, None, None, None, 0, 0, None) != 0
    assert lib.egx_allreduce(None, None, 16, 0, 1, None) != 0 and b"communicator" in lib.egx_last_error()
    assert lib.egx_comm_create(None, 0, 1, None) != 0
    assert lib.egx_comm_size(None) == -1 and lib.egx_comm_destroy(None) == 0
    assert lib.egx_comm_unique_id(None) != 0
    n += 11
    # > 2 GiB layouts come back intact
    big = Config(768, 8, 2048, 4, 4, 1e-5, 0, 0, 0.1, 0.0, 0.0)
    bs = segs_of([8192, 8192, 768, 2048], 32)
    bs[2].proj_w = 0
    assert lib.egx_encoder_workspace(C.byref(big), bs, 256, C.byref(sv), C.byref(sc)) == 0 and sv.value > 2 ** 32
    # decoder layouts
    for d, h, V, sy, S in ((256, 4, 7, 2, 45), (512, 8, 620, 4, 48), (256, 4, 7, 2, 450), (256, 4, 7, 9, 45), (100, 4, 7, 2, 45)):
        for compute in (0, 1):
            dc = DecConfig(d, h, 2048, 3, V, sy, S, 1e-5, compute, 0.1, 0.1, None)
            rc = lib.egx_decoder_workspace(C.byref(dc), 256, C.byref(sv), C.byref(sc))
            assert rc == 0 or lib.egx_last_error()
            n += 1
    assert lib.egx_ffn_dw_scratch(11520, 2048, 2) > 0 and lib.egx_linear_ce_scratch(3840, 128, 2) > 0
    assert lib.egx_wide_gemm_scratch(2, 768, 2048, 32768) > 0
    lib.egx_comm_library()
    lib.egx_launch_count(1)
    lib.egx_timing_enable(0)
    return n + 6


if __name__ == "__main__":
    count = exercise(bind(sys.argv[1]))
    print(f"host paths ok: {count} calls")
