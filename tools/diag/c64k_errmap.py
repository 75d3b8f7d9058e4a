# debug helper: where does conv_c64k differ from the chunked kernel?  (run via split_ab internals)
import os, sys
sys.argv = ["split_ab.py", "64", "64", "24", os.environ.get("N_IMG", "8")]
os.environ["ERRMAP"] = "1"
__file__ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "split_ab.py")
exec(open(__file__).read().split("flops = 2.0")[0])
import torch
run(3); torch.cuda.synchronize(); ref3 = out.clone()
run(1); torch.cuda.synchronize(); got = out.clone()
nanmask = ~torch.isfinite(got).reshape(-1, 64)
print('non-finite fraction', float(nanmask.float().mean()), 'by 128-px tile (first 40):', [round(float(x), 2) for x in nanmask.reshape(-1, 128, 64).float().mean(dim=(1, 2))[:40]])
got = torch.nan_to_num(got, nan=1e3, posinf=1e3, neginf=-1e3)
d = (got - ref3).abs().reshape(-1, 64)            # [pixel][channel]
npx = d.shape[0]
print("pixels", npx, "max", float(d.max()))
blk = d.reshape(npx // 32, 32, 2, 32).amax(dim=(1, 3))     # [32-px block][channel block]
print("by (pixel block % 4, cb):")
for b in range(4):
    print(b, [float(blk[b::4, c].max()) for c in range(2)])
bad = (d > 1e-3)
print("bad fraction", float(bad.float().mean()))
print("bad by pixel-in-block row r (of 32):", [float(bad.reshape(-1, 32, 64)[:, r, :].float().mean()) for r in range(32)])
print("bad by channel in block:", [round(float(bad.reshape(-1, 2, 32)[:, :, c].float().mean()), 2) for c in range(32)])
# residual / bias suspicion: difference vs residual
dd = (got - ref3).reshape(-1, 64)
rr = res.reshape(-1, 64)
print("corr of diff with residual:", float((dd * rr).sum() / (rr * rr).sum()))
