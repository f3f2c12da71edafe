"""Batched forwards for the insertion / deletion metric.

Mirrors (paths into /root/reference):
  gkern, auc                      evaluation.py:11-25, 41-43
  CausalMetric(...).evaluate      evaluate_test_dataset_milvus.py:32-85 (the variant the milvus
                                  evaluation runs; evaluation.py:46-138 `single_run` is its older twin)

The reference modifies `step` pixels, runs ONE B=1 forward, and repeats n_steps + 1 times (52 sequential
forwards per query-hit pair at step 1000 on 224x224), each of them ~190 kernel launches on a GPU that
is almost idle at B=1.  The images of all steps are known up front: pixel p changes after step
t(p) = rank of p in decreasing saliency // step, so image i is `where(t < i, finish, start)`.  This
class builds them on the device in one pass and pushes them through the embedder as large batches
(`max_batch`), so the whole curve costs a handful of full-occupancy forwards.  Same scores up to the
embedder's batch-size invariance (fp32, 1e-6), same return values.
"""
import numpy as np
import torch
import torch.nn.functional as F


def gkern(klen, nsig):
    """Gaussian blur kernel [3,3,klen,klen] (evaluation.py:11-25): a smoothed dirac per channel."""
    from scipy.ndimage import gaussian_filter
    inp = np.zeros((klen, klen))
    inp[klen // 2, klen // 2] = 1
    k = gaussian_filter(inp, nsig)
    kern = np.zeros((3, 3, klen, klen))
    for c in range(3):
        kern[c, c] = k
    return torch.from_numpy(kern.astype("float32"))


def auc(arr):
    """Normalised area under the curve (evaluation.py:41-43)."""
    return (arr.sum() - arr[0] / 2 - arr[-1] / 2) / (arr.shape[0] - 1)


class CausalMetric:
    def __init__(self, model, mode, step, substrate_fn, input_size=224, max_batch=256):
        assert mode in ["del", "ins"]
        self.model = model
        self.mode = mode
        self.step = step
        self.substrate_fn = substrate_fn
        self.hw = input_size * input_size
        self.max_batch = int(max_batch)

    @staticmethod
    def _embed(model, x):
        out = model(x)
        if isinstance(out, dict):
            out = out["embedding"]
        elif isinstance(out, tuple):
            out = out[0]
        return out

    def change_step(self, explanation, device):
        """t[p] = the step after which pixel p has been replaced: its rank in decreasing saliency
        (np.flip(np.argsort(...)), ties in that order) // step."""
        order = np.flip(np.argsort(np.asarray(explanation).flatten())).copy()
        rank = np.empty(self.hw, dtype=np.int64)
        rank[order] = np.arange(self.hw)
        return torch.from_numpy(rank // self.step).to(device)

    def evaluate(self, img_tensor, retrieved_tensor, explanation):
        """-> (auc, scores [n_steps + 1] float64, zero_counter) like the reference."""
        device = img_tensor.device
        n_steps = (self.hw + self.step - 1) // self.step
        side = int(self.hw ** 0.5)
        with torch.no_grad():
            q_feat = self._embed(self.model, img_tensor)
            if self.mode == "del":
                start, finish = retrieved_tensor.clone(), self.substrate_fn(retrieved_tensor)
            else:
                start, finish = self.substrate_fn(retrieved_tensor), retrieved_tensor.clone()
            start = start.reshape(1, 3, self.hw)
            finish = finish.reshape(1, 3, self.hw)
            t = self.change_step(explanation, device).view(1, 1, self.hw)
            sims = []
            for lo in range(0, n_steps + 1, self.max_batch):
                idx = torch.arange(lo, min(lo + self.max_batch, n_steps + 1), device=device).view(-1, 1, 1)
                imgs = torch.where(t < idx, finish, start).reshape(-1, 3, side, side)
                sims.append(F.cosine_similarity(q_feat, self._embed(self.model, imgs)))
            sims = torch.cat(sims).double().cpu().numpy()
        zero_counter = int(np.count_nonzero(sims < 0))
        scores = np.where(sims < 0, 0.0, sims)          # the reference clamps only negative values
        return auc(scores), scores, zero_counter
