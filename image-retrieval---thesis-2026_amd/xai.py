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


# ---- explanations.py:15-152 (SBSM / SBSMBatch: sliding-window occlusion saliency) ---------------------------
def sliding_window_masks(input_size, window_size, stride):
    """explanations.py:36-63: uint8 [N, 1, H, W], 1 outside the window, 0 inside; windows start at
    stride - window_size and step by stride (clipped at the borders)."""
    h, w = input_size
    rows = np.arange(stride - window_size, h, stride)
    cols = np.arange(stride - window_size, w, stride)
    masks = np.ones((len(rows) * len(cols), h, w), dtype=np.uint8)
    i = 0
    for r in rows:
        for c in cols:
            masks[i, max(r, 0):min(r + window_size, h), max(c, 0):min(c + window_size, w)] = 0
            i += 1
    return masks.reshape(-1, 1, h, w)


class SBSMBatch:
    """Same constructor / generate_masks / load_masks / call as explanations.py:15-152 (`SBSMBatch(model,
    input_size, gpu_batch)`, `explainer(x_q, x)` or `explainer(x)` for self-similarity) -> saliency [B, H, W].

    MI355X design: the reference materialises all B * N masked images ([B*N, C, H, W], 0.6 MB each at 224x224)
    and then an [H, W, B, N] tensor `K` that it sums over N.  Here the masked images exist only one
    `gpu_batch` chunk at a time (built on the device from the uint8 masks), and the saliency is the matrix
    product  sal[b] = (1 - masks)^T [HW x N] . gain[b] [N] / count  -- K never exists.  Distances are
    Euclidean on the embedder's outputs like `torch.cdist` / `torch.norm` there."""

    def __init__(self, model, input_size, gpu_batch=100):
        self.model = model
        self.input_size = tuple(input_size)
        self.gpu_batch = int(gpu_batch)
        self.masks = None

    def _set_masks(self, masks, device=None):
        dev = device if device is not None else next(self.model.parameters()).device
        self.masks = torch.from_numpy(np.ascontiguousarray(masks)).to(dev)
        self.N = self.masks.shape[0]
        inv = (1 - self.masks.reshape(self.N, -1)).float()                  # [N, HW]: 1 inside the window
        self._inv_t = inv.t().contiguous()                                  # [HW, N]
        self._count = inv.sum(dim=0)                                        # windows covering each pixel

    def generate_masks(self, window_size, stride, savepath="masks.npy"):
        masks = sliding_window_masks(self.input_size, window_size, stride)
        if savepath:
            np.save(savepath, masks)
        self._set_masks(masks)
        self.window_size, self.stride = window_size, stride

    def load_masks(self, filepath):
        self._set_masks(np.load(filepath))

    def _embed_masked(self, x):
        """Embeddings of mask n applied to image b, n-major like the reference's stack: row n * B + b."""
        b, c, h, w = x.shape
        out = []
        per = max(1, self.gpu_batch // b)                                   # masks per chunk
        for n0 in range(0, self.N, per):
            m = self.masks[n0:n0 + per].to(x.dtype)                         # [n, 1, H, W]
            chunk = (m[:, None] * x[None]).reshape(-1, c, h, w)             # [n * B, C, H, W]
            out.append(CausalMetric._embed(self.model, chunk))
        return torch.cat(out)

    def __call__(self, x_q, x=None):
        return self.forward(x_q, x)

    def forward(self, x_q, x=None):
        self_sim = x is None
        if self_sim:
            x = x_q
        b = x.shape[0]
        h, w = self.input_size
        with torch.no_grad():
            e_q = CausalMetric._embed(self.model, x_q)
            e_m = self._embed_masked(x).reshape(self.N, b, -1)              # [N, B, D]
            if self_sim:
                gain = torch.linalg.vector_norm(e_q[None] - e_m, dim=2).t()                 # [B, N]
            else:
                e_r = CausalMetric._embed(self.model, x)
                o_dist = torch.cdist(e_q, e_r).reshape(-1, 1)                               # [Q * B, 1]
                m_dist = torch.cdist(e_q, e_m.reshape(self.N * b, -1))                      # [Q, N * B]
                m_dist = m_dist.reshape(-1, self.N, b).permute(0, 2, 1).reshape(-1, self.N)
                gain = (m_dist - o_dist).clamp(min=0)                                       # [Q * B, N]
            sal = (gain.float() @ self._inv_t.t()) / self._count                            # [., HW]
        return sal.reshape(-1, h, w)
