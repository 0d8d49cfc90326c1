"""opt.normal: the Yeo-Johnson power transform around the sampler (Model/model.py:762-808; call sites
Utils/train_test_utils.py:560-562,578-588 and Model/model.py:616-617).  Host-side scikit-learn, exactly as in the
reference (PowerTransformer(method='yeo-johnson'), standardised) -- off in every shipped configuration.

Per-slice semantics, as everywhere in this build: every slice of a batch gets its own transformer (the reference fits
one over the whole batch tensor; at B = 1 the two are the same)."""
import numpy as np
import torch


class SliceTransformers(list):
    """One fitted sklearn PowerTransformer per slice, in batch order."""


def yeo_johnson_transform(img_tensor):
    """[B, 1, H, W] -> (transformed tensor on the same device / dtype float64 as sklearn returns, transformers)."""
    from sklearn.preprocessing import PowerTransformer
    x = img_tensor.detach().cpu().numpy()
    out = np.empty(x.shape, dtype=np.float64)
    trs = SliceTransformers()
    for b in range(x.shape[0]):
        tr = PowerTransformer(method="yeo-johnson")
        out[b] = tr.fit_transform(x[b].reshape(-1, 1)).reshape(x[b].shape)
        trs.append(tr)
    return torch.from_numpy(out).to(img_tensor.device), trs


def yeo_johnson_inverse_transform(transformed_img_tensor, transformer):
    x = transformed_img_tensor.detach().cpu().numpy()
    if not isinstance(transformer, SliceTransformers):          # a bare sklearn transformer: the reference's form
        transformer = SliceTransformers([transformer] * x.shape[0])
    out = np.empty(x.shape, dtype=x.dtype)
    for b in range(x.shape[0]):
        out[b] = transformer[b].inverse_transform(x[b].reshape(-1, 1)).reshape(x[b].shape)
    return torch.from_numpy(out).to(transformed_img_tensor.device)
