"""Where a MilvusRetriever.search call spends its time (development aid): cProfile over 20 single-image queries."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx.model import DenseNet121                      # noqa: E402
from mirx.retriever import MilvusManager, MilvusRetriever, default_transform    # noqa: E402


def main():
    from PIL import Image
    dev = torch.device("cuda:0")
    model = DenseNet121().eval().to(dev)
    mgr = MilvusManager(device=dev)
    mgr.connect()
    mgr.create_collection("densenet121", drop_old=True)
    col = mgr.collections["densenet121"]
    g = torch.Generator(device=dev).manual_seed(7)
    for s0 in range(0, 200_000, 50_000):
        emb = torch.nn.functional.normalize(torch.randn((50_000, 1024), generator=g, device=dev), dim=1)
        col.insert([[f"img_{s0 + i}.png" for i in range(50_000)], ["normal"] * 50_000, emb])
    r = MilvusRetriever(mgr, "densenet121", model, default_transform(224))
    r.load_collection()
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (300, 280, 3), dtype=np.uint8))
    for _ in range(3):
        r.search(img, top_k=10)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        r.search(img, top_k=10)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
