import os, sys, torch
sys.path.insert(0, os.getcwd())
from oracle.golden_inputs import g5_inputs
from oracle import xfm_oracle as O
from tests.helpers import load_json, load_npz
from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
from xfmamba_amd import fusion_vmamba as fv, ss2d as S2, csms6s
shapes = load_json("g5_state_shapes.json")["tiny"]; z = load_npz("g5_model.npz")
m = TwoViewXFMambaTop(1, 2, type="tiny"); m.load_state_dict(O.synth_state_dict(shapes, 0)); m = m.cuda().eval()
xa, xb, _ = (t.cuda() for t in g5_inputs())
ref = torch.from_numpy(z["logits_eval"])
def run(tag):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(xa, xb)
    print(tag, float((y.float().cpu()-ref).abs().max()/ref.abs().max()), y.float().cpu().tolist())
run("bf16 default")
# variant: scan operands in fp32
orig = S2.ss2d_core_fn
S2f = lambda x, dts, A, Bs, Cs, D, b, H, W: orig(x.float(), dts.float(), A, Bs.float(), Cs.float(), D, b, H, W)
fv.ss2d_core_fn = S2f
run("scan operands upcast (still bf16-rounded)")
fv.ss2d_core_fn = orig
# variant: projections x_proj/dt_proj in fp32 (matmul outside autocast)
orig_core = fv._ss2d_core
def core32(x, *a, **k):
    with torch.autocast("cuda", enabled=False):
        return orig_core(x.float(), *a, **k)
fv._ss2d_core = core32
run("x_proj/dt_proj/scan in fp32")
fv._ss2d_core = orig_core
with torch.no_grad():
    y = m(xa, xb)
print("fp32", float((y.cpu()-ref).abs().max()/ref.abs().max()))
