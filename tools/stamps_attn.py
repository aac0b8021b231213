"""Phase stamps of the attention-side forward launch of the cut mode (fused_fwd_kernel<..., CUT>; needs a -DEGX_STAMPS build, EGX_LIB=egot2_amd/_variants/lib_stamps.so):
workgroup 0, thread 0, s_memtime cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EGX_FFN_CUT"] = "1"
import torch
from egot2_amd import hhi_ttm, _lib
import egot2_amd.functional as _F_tuning; _F_tuning.reload_tuning_each_call = True   # the switches below are flipped inside this process
from egot2_amd.synth import hhi_args
lib = _lib.load()
dev = torch.device("cuda:0")
m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5)).to(dev).train()
feats = [torch.randn(256, 15, 256, device=dev) for _ in range(3)]
lib.egx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for comp in ("f32s", "bf16", "f32"):
    m.set_compute(comp, "fused")
    with torch.no_grad():
        for _ in range(3):
            m.forward_features(*feats)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps(buf, 32)
    b = list(buf)
    order = [(0, "start"), (10, "segtab"), (1, "token-prep GEMM"), (2, "LN0+emb"), (3, "QKV"), (4, "attention"), (5, "out-proj"), (6, "LN1+stores")]
    prev = b[0]
    parts = []
    for idx, name in order[1:]:
        parts.append(f"{name}={b[idx] - prev}")
        prev = b[idx]
    print(comp, "fused_fwd<CUT> to LN1:", b[6] - b[0], " ".join(parts))
    print("    cold start: kernel entry -> segtab barrier %d -> issue begins %d -> 28 loads issued %d -> LDS zero fill %d -> barrier %d" % (b[10] - b[0], b[12] - b[10], b[13] - b[12], b[14] - b[13], b[15] - b[14]))
    print("    token prep (wave 0): setup(before first step)=%d first-fragment-wait=%d mfma+refill=%d" % (b[21], b[22], b[23]))

# ---- attention-side backward launch of the cut mode (fused_bwd_kernel<..., CUT>)
for comp in ("f32s", "bf16", "f32"):
    m.set_compute(comp, "fused")
    for _ in range(3):
        for q_ in m.parameters():
            q_.grad = None
        m.forward_features(*feats).sum().backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.egx_debug_stamps(buf, -32)
    b = list(buf)
    order = [(12, "entry"), (4, "blocks in LDS"), (5, "LN1 bwd + g1 store"), (6, "colsums + out-proj dX"), (7, "QKV rows -> LDS"), (9, "attention bwd"),
             (10, "colsums + in-proj dX"), (11, "token-prep bwd")]
    prev = b[12]
    parts = []
    for idx, name in order[1:]:
        parts.append(f"{name}={b[idx] - prev}")
        prev = b[idx]
    print(comp, "fused_bwd<CUT>:", b[11] - b[12], " ".join(parts))
    print("    LN1 bwd (from phase start %d): request+hook %d | get %d | math %d | put %d | to barrier %d | barrier + g1 store %d" %
          (b[16] - b[4], b[17] - b[16], b[18] - b[17], b[19] - b[18], b[20] - b[19], b[21] - b[20], b[5] - b[21]))
    print("    LN0 bwd (from phase start %d): request+hook %d | get %d | math %d | put (incl. dseg stores) %d | to barrier %d | colsums %d" %
          (b[22] - b[10], b[23] - b[22], b[24] - b[23], b[25] - b[24], b[26] - b[25], b[27] - b[26], b[11] - b[27]))
