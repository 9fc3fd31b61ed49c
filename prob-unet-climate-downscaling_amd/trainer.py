"""Host mirror of the reference's training-loop call pattern around the hot path.

What the reference does per batch (train_prob_unet_model.py:105-158): `model.elbo(inputs, targets, timestamps, M=ensemble_size)`
-> 3-tuple `(loss, [recon_float], kl[B])`, then `optimizer.zero_grad(); loss.backward(); optimizer.step()`, and it
pulls THREE python floats per step off the device (`recon_list[0]`, `kl_div.mean().item()`, `loss.item()` for the progress
bar).  Here the same sequence runs against the HIP engine, with two differences that do not change results:
  * the per-step scalars stay on the device (`model.sync_scalars=False`) and are averaged in one stacked reduction at the
    end of the epoch -> one host sync per epoch instead of three per step;
  * any optimizer works (`torch.optim.AdamW(model.parameters())` as in main.py:103, or the fused `FlatAdamW`).
The epoch-level beta schedule of main.py:108-155 is `beta_schedule()`.

Batches are dicts with 'inputs' [B,Cin,H,W], 'targets' [B,Cout,H,W] and (optionally) 'timestamps' [B] — the keys of
ClimExDataset.__getitem__ (climex_utils.py:218-224); any iterable of such dicts serves as the dataloader.
"""
from typing import Iterable, Optional, Tuple

import torch


def beta_schedule(epoch: int, num_epochs: int, warmup_epochs: int = 2, max_beta_1: float = 1.0) -> Tuple[float, float]:
    """(beta_0, beta_1) to use for the epoch AFTER `epoch` has finished (main.py:141-155): no KL penalty during the warm-up
    epochs, then beta_1 ramps linearly to max_beta_1 at the last epoch.  Epochs count from 1; epoch 0 gives the initial
    values of main.py:109-110."""
    if epoch <= warmup_epochs:
        return 1.0, 0.0
    total = num_epochs - warmup_epochs
    progress = min((epoch - warmup_epochs) / total, 1.0) if total > 0 else 1.0
    return 1.0, progress * max_beta_1


def _batch_tensors(batch, device):
    inputs = batch["inputs"].to(device, non_blocking=True)
    targets = batch["targets"].to(device, non_blocking=True)
    ts = batch.get("timestamps") if hasattr(batch, "get") else None
    if ts is not None:
        ts = ts.unsqueeze(dim=1).to(device, non_blocking=True)      # shaped like the reference; the model ignores it
    return inputs, targets, ts


def _epoch_means(recon_vals, kl_vals):
    if not recon_vals:
        return float("nan"), float("nan")
    r = torch.stack([v if torch.is_tensor(v) else torch.tensor(float(v)) for v in recon_vals]).float().mean()
    k = torch.stack(kl_vals).float().mean()
    both = torch.stack([r.to(k.device), k]).cpu()                   # the epoch's only device->host sync
    return float(both[0]), float(both[1])


def train_probunet_step(model, dataloader: Iterable, optimizer, epoch: int = 1, num_epochs: int = 1, device=None,
                        ensemble_size: int = 5, progress: Optional[callable] = None) -> Tuple[float, float]:
    """One training epoch (train_prob_unet_model.py:105-158).  Returns (mean recon loss, mean KL) over the epoch."""
    model.train()
    device = device if device is not None else next(model.parameters()).device
    prev_sync = getattr(model, "sync_scalars", True)
    model.sync_scalars = False
    recon_vals, kl_vals = [], []
    try:
        for i, batch in enumerate(dataloader):
            inputs, targets, ts = _batch_tensors(batch, device)
            out = model.elbo(inputs, targets, ts, M=ensemble_size)
            loss, recon_list, kl_div = out[0], out[1], out[2]
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            recon_vals.append(recon_list[0])
            kl_vals.append(kl_div.mean())
            if progress is not None:
                progress(epoch, num_epochs, i, loss)                # receives the device scalar; .item() it only if displayed
    finally:
        model.sync_scalars = prev_sync
    return _epoch_means(recon_vals, kl_vals)


@torch.no_grad()
def eval_probunet_model(model, dataloader: Iterable, device=None, ensemble_size: int = 5) -> Tuple[float, float]:
    """Validation pass (train_prob_unet_model.py:161-210): ELBO forward only, eval mode (dropout off)."""
    model.eval()
    device = device if device is not None else next(model.parameters()).device
    prev_sync = getattr(model, "sync_scalars", True)
    model.sync_scalars = False
    recon_vals, kl_vals = [], []
    try:
        for batch in dataloader:
            inputs, targets, ts = _batch_tensors(batch, device)
            out = model.elbo(inputs, targets, ts, M=ensemble_size)
            recon_vals.append(out[1][0])
            kl_vals.append(out[2].mean())
    finally:
        model.sync_scalars = prev_sync
    return _epoch_means(recon_vals, kl_vals)


@torch.no_grad()
def sample_probunet_model(model, batch, device=None, num_samples: int = 3, residual_std=None, epsilon: float = 1e-10):
    """`num_samples` prior samples per input, reconstructed to physical units on the device
    (train_prob_unet_model.py:213-262 draws them one `model(x, training=False)` call at a time and converts each on the
    host with ClimExDataset.residual_to_hr).  Returns hr_preds [B, num_samples, C, H, W]; with residual_std=None the
    standardised residuals are returned (sample_residual_probunet_model, :265-300)."""
    model.eval()
    device = device if device is not None else next(model.parameters()).device
    inputs = batch["inputs"].to(device)
    if residual_std is None:
        return model.sample(inputs, num_samples)
    lrinterp = batch["lrinterp"].to(device)
    return model.sample_hr(inputs, num_samples, lrinterp, residual_std.to(device), epsilon=epsilon)


def fit(model, train_loader, val_loader, optimizer, num_epochs: int, device=None, ensemble_size: int = 5,
        warmup_epochs: int = 2, max_beta_1: float = 1.0, on_epoch: Optional[callable] = None):
    """The epoch loop of main.py:126-175 (betas set before each epoch, annealed after it).  Returns the four history lists
    (train recon, train KL, val recon, val KL)."""
    tr, tk, vr, vk = [], [], [], []
    beta_0, beta_1 = beta_schedule(0, num_epochs, warmup_epochs, max_beta_1)
    for epoch in range(1, num_epochs + 1):
        model.beta_0, model.beta_1 = beta_0, beta_1
        a, b = train_probunet_step(model, train_loader, optimizer, epoch, num_epochs, device, ensemble_size)
        tr.append(a); tk.append(b)
        beta_0, beta_1 = beta_schedule(epoch, num_epochs, warmup_epochs, max_beta_1)
        if val_loader is not None:
            a, b = eval_probunet_model(model, val_loader, device, ensemble_size)
            vr.append(a); vk.append(b)
        if on_epoch is not None:
            on_epoch(epoch, tr[-1], tk[-1], vr[-1] if vr else None, vk[-1] if vk else None)
    return tr, tk, vr, vk
