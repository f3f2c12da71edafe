"""Sequential CPU restatement of the reference's insertion / deletion loop (test infrastructure only).

Follows evaluate_test_dataset_milvus.py:43-85 of /root/reference: one B=1 forward per step, pixels
swapped in place between forwards.  `model` is any callable returning [1, D] embeddings (the tests
pass the CPU DenseNet restatement).  Parity: unpinned by reference fixtures (the reference has none for
this loop); the loop itself is plain indexing, the model is the oracle's.
"""
import numpy as np
import torch
import torch.nn.functional as F


def auc(arr):
    return (arr.sum() - arr[0] / 2 - arr[-1] / 2) / (arr.shape[0] - 1)


def causal_metric(model, mode, step, substrate_fn, img, retrieved, explanation):
    hw = img.shape[-1] * img.shape[-2]
    side = img.shape[-1]
    n_steps = (hw + step - 1) // step
    with torch.no_grad():
        q = model(img)
        if mode == "del":
            start, finish = retrieved.clone(), substrate_fn(retrieved)
        else:
            start, finish = substrate_fn(retrieved), retrieved.clone()
        start = start.reshape(1, 3, hw)
        finish = finish.reshape(1, 3, hw)
        order = torch.from_numpy(np.flip(np.argsort(explanation.flatten())).copy())
        scores = np.empty(n_steps + 1)
        zeros = 0
        for i in range(n_steps + 1):
            r = model(start.reshape(1, 3, side, side))
            c = float(F.cosine_similarity(q, r)[0])
            if c < 0:
                c = 0.0
                zeros += 1
            scores[i] = c
            if i < n_steps:
                coords = order[step * i: step * (i + 1)]
                start[0, :, coords] = finish[0, :, coords]
    return auc(scores), scores, zeros


# -- explanations.py:105-152 (SBSMBatch.forward), the tensors of the reference kept as they are there -------
def sbsm_batch(model, masks, x_q, x=None, gpu_batch=100):
    """masks: uint8 [N, 1, H, W] (explanations.py:36-63); returns saliency [B, H, W] (float32)."""
    masks = torch.from_numpy(np.asarray(masks))
    n = masks.shape[0]
    self_sim = x is None
    if self_sim:
        x = x_q
    b, c, h, w = x.shape
    e_q = model(x_q)
    if not self_sim:
        o_dist = torch.cdist(e_q, model(x)).view(-1, 1)
    stack = torch.mul(masks.view(n, 1, h, w), x.view(b * c, h, w)).view(b * n, c, h, w)
    e_m = torch.cat([model(stack[i:min(i + gpu_batch, n * b)]) for i in range(0, n * b, gpu_batch)])
    if self_sim:
        m_dist = torch.norm(e_q.unsqueeze(1) - e_m.view(-1, b, e_q.shape[1]).permute(1, 0, 2), dim=2)
        k = (1 - masks).permute(2, 3, 1, 0) * m_dist
    else:
        m_dist = torch.cdist(e_q, e_m).view(-1, n, b).permute(0, 2, 1).reshape(-1, n)
        k = (1 - masks).permute(2, 3, 1, 0) * (m_dist - o_dist).clamp(min=0)
    count = n - masks.sum(dim=(0, 1))
    return (k.sum(dim=-1).permute(2, 0, 1) / count).float()
