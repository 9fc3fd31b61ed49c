"""End-to-end demonstration on one MI355X: synthetic ClimEx-shaped hi-res fields -> device item transform (row f4) -> the trainer
loop with the beta schedule (row f1) on the cfg3 network (4 -> 1 planes, 256 x 256, depth 5) -> prior samples reconstructed to
physical units (row f3).  Writes profiles/r1_training_demo.json (per-epoch means, pairs/s).  Usage: python tools/train_demo.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import probunet_amd as pa

dev = torch.device("cuda", 0)
torch.manual_seed(42)
N, B, EPOCHS = 256, 32, 8
MAX_BETA_1 = 0.01          # KL(q||p) starts at ~4e5 on these synthetic fields: with Adam, a full-weight KL term switched on after the warm-up
                           # throws the prior's log-sigma head far past its optimum (observed: sigma_p -> 1e21 -> inf); the reference's
                           # schedule (main.py:108-155) is kept, scaled by max_beta_1
g = torch.Generator(device="cpu").manual_seed(0)
# smooth large-scale structure + fine-scale detail correlated with it: the detail is what the network has to learn
lo = torch.randn(N, 4, 16, 16, generator=g)
big = torch.nn.functional.interpolate(lo, scale_factor=16.0, mode="bicubic", align_corners=False)
fine = torch.nn.functional.interpolate(torch.randn(N, 4, 64, 64, generator=g), scale_factor=4.0, mode="bilinear", align_corners=False)
hr = (big + 0.35 * fine * (1.0 + 0.5 * torch.tanh(big))).to(dev)
tr = pa.data.ClimExTransform(lowres_scale=4).fit(hr)
model = pa.ProbabilisticUNet(4, 1, 12, [32, 64, 128, 256, 512], 32, [1, 2, 4, 8, 16], 1.0, 0.0, 0.0, dtype="f16", max_batch=B, max_members=5).to(dev)
opt = pa.FlatAdamW(model, lr=1e-4)


def batches(lo_i, hi_i):
    out = []
    for i in range(lo_i, hi_i, B):
        it = tr(hr[i:i + B])
        out.append({"inputs": it["inputs"], "targets": it["targets"][:, :1].contiguous(), "lrinterp": it["lrinterp"], "hr": it["hr"]})
    return out


train, val = batches(0, N - B), batches(N - B, N)
hist = []
t0 = time.perf_counter()
def on_epoch(ep, tr_r, tr_k, va_r, va_k):
    torch.cuda.synchronize()
    hist.append(dict(epoch=ep, beta_1=float(model.beta_1), train_afcrps=tr_r, train_kl=tr_k, val_afcrps=va_r, val_kl=va_k, seconds=round(time.perf_counter() - t0, 2)))
    print(hist[-1], flush=True)
pa.trainer.fit(model, train, val, opt, num_epochs=EPOCHS, device=dev, ensemble_size=5, warmup_epochs=2, max_beta_1=MAX_BETA_1, on_epoch=on_epoch)
el = time.perf_counter() - t0
with torch.no_grad():
    std1 = tr.lrstats[1][1][:1]
    hrp = pa.trainer.sample_probunet_model(model, val[0], dev, num_samples=8, residual_std=std1)                   # prior samples
    hrq = model.sample_hr(val[0]["inputs"], 8, val[0]["lrinterp"], std1, target=val[0]["targets"])                 # posterior samples
    truth = val[0]["hr"][:, :1]
    base = float((val[0]["lrinterp"][:, :1] - truth).abs().mean())
    ens = float((hrp.mean(1) - truth).abs().mean())
    ens_q = float((hrq.mean(1) - truth).abs().mean())
out = dict(config="cfg3 network, f16 engine, afCRPS M=5, B=32, FlatAdamW lr 1e-4, max_beta_1 %g, synthetic fields, %d train / %d val pairs" % (MAX_BETA_1, N - B, B),
           history=hist, train_pairs_per_s=round(EPOCHS * (N - B) / el, 1), wall_s=round(el, 2),
           mae_lrinterp_vs_truth=base, mae_prior_ensemble_mean_vs_truth=ens, mae_posterior_ensemble_mean_vs_truth=ens_q,
           finite=bool(torch.isfinite(hrp).all() and torch.isfinite(hrq).all()))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r1_training_demo.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "history"}))
