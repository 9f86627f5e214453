"""Experiment: two contexts on CU-masked streams (hipExtStreamCreateWithCUMask), each simulating its own tiles on a share of the
CUs at the same time, against one context on the whole device.  A tile's kernels alternate between issue-bound (k_sample, k_depth)
and traffic-bound (k_gl at low depth) phases; two tiles out of phase on disjoint CUs can overlap them, which two kernels on one
set of CUs do not (the dispatcher runs such grids one after the other).
usage (GPU box): python tools/cu_split.py [c5|c5u8|fixedq|c3] [share of stream A in 1/8ths, default 4]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
shareA = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = {"c5": (500, dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1), True, ["fmt_dp", "gl", "pl"]),
     "c5u8": (500, dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1), True, ["fmt_dp", "gl", "pl_u8"]),
     "fixedq": (1000, dict(depth=20.0, error_rate=0.01, gl_model=2), False, ["fmt_dp", "gl"]),
     "c3": (1000, dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2), False, ["fmt_dp", "gl"])}
N, kw, homref, fields = W[name]
S, TILES = 65536, 8
hip = C.CDLL("libamdhip64.so")
props_cus = torch.cuda.get_device_properties(0).multi_processor_count
nwords = (props_cus + 31) // 32
def masked_stream(pred):
    m = (C.c_uint32 * nwords)()
    for cu in range(props_cus):
        if pred(cu): m[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), nwords, m)
    assert rc == 0, rc
    return s
gt = torch.zeros((S, N), dtype=torch.uint8, device="cuda:0") if homref else synth.binary_sites_torch(0, S, N, "cuda:0")
def run(streams, label):
    sims = [Simulator(VcfglArgs(seed=42, **kw), N, max_sites_per_tile=S) for _ in streams]
    tiles = [s.new_tile(S, fields=fields, device="cuda:0") for s in sims]
    for s, t, st in zip(sims, tiles, streams):
        s.simulate_device(0, gt, t, stream=st.value if st else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(TILES):
        i = k % len(sims)
        sims[i].simulate_device(k * S, gt, tiles[i], stream=streams[i].value if streams[i] else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in sims: s.check(); s.close()
    print(f"{name} {label}: {TILES} tiles in {dt * 1e3:.2f} ms = {dt / TILES * 1e3:.3f} ms per tile, {TILES * S * N / dt:.3e} evals/s", flush=True)
print("CUs", props_cus)
run([None], "one context, whole device")
run([masked_stream(lambda cu: True)], "one context, stream with a full CU mask")
# a share of every group of 8 consecutive CUs to stream A, the rest to stream B
run([masked_stream(lambda cu: cu % 8 < shareA), masked_stream(lambda cu: cu % 8 >= shareA)], f"two contexts, CU % 8 < {shareA} | >= {shareA}")
half = props_cus // 2
run([masked_stream(lambda cu: cu < half), masked_stream(lambda cu: cu >= half)], "two contexts, lower | upper half of the CU indices")
run([None, None], "two contexts, both on the null stream (no overlap possible)")
