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
