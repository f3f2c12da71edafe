"""evaluate(model, loader, device, args): drop-in for the reference's hot loop
(test.py:1065-1126; twin test_nonclip.py:132-172).

Same stages, prints and .npz fields -- embed loop -> scores -> self-exclusion -> R@K -> full
ranking -> mAP / mP@K -> majority-vote classification -> np.savez -- but the N x N distance
matrix, topk and argsort of the reference are replaced by one device-resident index and
libmirx's exact full ranking (fp64 scores, ties -> lowest index).
"""
import os

import numpy as np
import torch

from .index import FlatIndex
from .metrics import compute_classification_metrics, compute_map, compute_map_multilabel, retrieval_accuracy


@torch.no_grad()
def embed_loader(model, loader, device):
    """test.py:1070-1078: forward every batch, concatenate embeddings and labels on `device`."""
    model.eval()
    embeds, labels = [], []
    for data in loader:
        out = model(data[0].to(device))
        if isinstance(out, dict):
            out = out["embedding"]
        elif isinstance(out, tuple):
            out = out[0]
        embeds.append(out)
        labels.append(data[1].to(device))
    return torch.cat(embeds, dim=0), torch.cat(labels, dim=0)


def rank_self(embeds, metric="cdist"):
    """Self-retrieval ranking of a [N,D] embedding set on its GPU.
    -> (ranks [N,N] int64 row-per-query, self last; reported scores [N,N] in rank order)."""
    n, d = embeds.shape
    if n > 65536:
        raise ValueError("evaluate(): full ranking is limited to 65536 images (the reference needs an "
                         "N x N matrix and stops far earlier); use FlatIndex.search for top-k")
    ix = FlatIndex(d, "L2" if metric in ("cdist", "L2", "l2") else "COSINE", embeds.device.index or 0)
    ix.add(embeds)
    me = torch.arange(n, device=embeds.device)
    return ix.rank_all(embeds, exclude_ids=me, with_scores=True)


@torch.no_grad()
def evaluate(model, loader, device, args):
    embeds, labels = embed_loader(model, loader, device)
    metric = getattr(args, "metric", "cdist")
    ranks, sorted_scores = rank_self(embeds.float(), metric)
    labels_np = labels.cpu().numpy()
    k_values = [1, 5, 10, 15, 20]
    # only the heads of the lists leave the GPU; AP / precision@k run over the full ranking on the
    # device (mirx_rank_metrics) -- no N x N matrix crosses PCIe
    ranks_head = ranks[:, :max(k_values)].cpu().numpy()

    kappas = [1, 5, 10]
    accuracy = retrieval_accuracy(None, labels_np, topk=kappas, topk_ids=ranks_head[:, :max(kappas)])
    accuracy = torch.stack(accuracy).cpu().numpy()
    print(">> R@K{}: {}%".format(kappas, np.around(accuracy, 2)))

    mAP, _, pr, _ = compute_map(ranks.t(), labels_np, kappas)
    print(">> mAP: {:.2f}%".format(mAP * 100.0))
    print(">> mP@K{}: {}%".format(kappas, np.around(pr * 100.0, 2)))

    print("\n>> Classification Metrics (Majority Voting):")
    classification_results = compute_classification_metrics(labels_np, None, k_values, ranks=ranks_head.T)
    for k in k_values:
        m = classification_results[k]
        print(f"\n>> Top-{k} Retrieved Images:")
        print(f'   Accuracy: {m["accuracy"]:.2f}%')
        print(f'   Precision (macro): {m["precision_macro"]:.2f}%')
        print(f'   Recall (macro): {m["recall_macro"]:.2f}%')
        print(f'   F1 (macro): {m["f1_macro"]:.2f}%')
        print(f'   Precision (weighted): {m["precision_weighted"]:.2f}%')
        print(f'   Recall (weighted): {m["recall_weighted"]:.2f}%')
        print(f'   F1 (weighted): {m["f1_weighted"]:.2f}%')

    result = {"embeds": embeds, "labels": labels, "ranks": ranks, "acc": accuracy, "mAP": mAP, "pr": pr,
              "classification": classification_results}
    if getattr(args, "save_dir", None):
        os.makedirs(args.save_dir, exist_ok=True)
        file_name = args.resume.split("/")[-1].split(".")[0]
        save_path = os.path.join(args.save_dir, file_name)
        # test.py:1124 saves dists = -(-cdist) = +L2 with a +inf diagonal (cosine: -similarity)
        n = ranks.shape[0]
        dists = torch.empty((n, n), dtype=torch.float32, device=ranks.device)
        dists.scatter_(1, ranks, -sorted_scores)
        dists.fill_diagonal_(float("inf"))
        np.savez(save_path, embeds=embeds.cpu().numpy(), labels=labels_np, dists=dists.cpu().numpy(),
                 kappas=kappas, acc=accuracy, mAP=mAP, pr=pr,
                 classification_k_values=list(classification_results.keys()),
                 **{f"classification_k{k}": np.array(list(v.values())) for k, v in classification_results.items()})
    return result


@torch.no_grad()
def evaluate_multilabels(model, loader, device, args):
    """test.py:987-1062 (VinDr-CXR style multi-hot labels): embed loop -> cosine similarities with the diagonal excluded
    -> mAP at Jaccard > 0.25 and > 0.5 (compute_map_multilabel) -> Precision@K (share of the top K with at least one label
    in common with the query) and Recall@K (1 if any of the top K shares a label) for K in 1, 5, 10, 15, 20 -> optional
    np.savez(embeds, labels).  Same prints; returns the numbers (the reference returns None).  The N x N matrix, its
    two argsorts and the Python double loop are replaced by one resident index, one exact full ranking and tensor ops on
    the device."""
    embeds, labels = embed_loader(model, loader, device)
    embeds_norm = torch.nn.functional.normalize(embeds.float(), p=2, dim=1)
    ranks, _ = rank_self(embeds_norm, "cosine")                       # [N, N] row per query, self last
    print("\n--- VinDr-CXR Retrieval Results ---")
    out = {"mAP": {}, "precision": {}, "recall": {}}
    for t in [0.25, 0.5]:
        m = compute_map_multilabel(None, labels, threshold=t, ranks=ranks.t())
        out["mAP"][t] = float(m)
        print(f">> mAP (Jaccard > {t}): {m * 100.0:.2f}%")
    k_values = [1, 5, 10, 15, 20]
    lab = labels.to(ranks.device).float()
    top = ranks[:, :max(k_values)]
    matches = (lab[top] * lab[:, None, :]).sum(dim=2) > 0              # [N, maxk]: shares a label with the query
    print(f'\n{"K":<5} | {"Precision@K":<15} | {"Recall@K":<15}')
    print("-" * 40)
    n = lab.shape[0]
    for k in k_values:
        cnt = matches[:, :k].sum(dim=1)
        avg_precision = float((cnt.double() / k).sum() / n * 100)
        avg_recall = float((cnt > 0).double().sum() / n * 100)
        out["precision"][k], out["recall"][k] = avg_precision, avg_recall
        print(f"{k:<5} | {avg_precision:<15.2f}% | {avg_recall:<15.2f}%")
    if getattr(args, "save_dir", None):
        os.makedirs(args.save_dir, exist_ok=True)
        save_path = os.path.join(args.save_dir, "evaluation_results.npz")
        np.savez(save_path, embeds=embeds.cpu().numpy(), labels=labels.cpu().numpy())
        print(f"\n>> Results saved to {save_path}")
    return out
