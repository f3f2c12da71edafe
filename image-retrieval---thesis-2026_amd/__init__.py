"""mirx -- MI355X-native embedding + exhaustive retrieval path (drop-in for the hot loop of
CrispyChillies/Image-Retrieval---Thesis-2026: test.py evaluate(), MilvusRetriever.search()).

Layout
  csrc/        HIP kernels + the C ABI (include/mirx.h) -> libmirx.so (built in-tree)
  _lib.py      ctypes binding of libmirx.so (fails loudly when the library is missing)
  index.py     FlatIndex: device-resident gallery, exact top-k / full ranking
  model.py     DenseNet121 / ... embedders with the reference's forward() contract
  retriever.py MilvusManager / MilvusRetriever-shaped in-process retrieval
  metrics.py   retrieval_accuracy, compute_map, precision_at_k, ... (reference names)
  evaluate.py  evaluate(model, loader, device, args) drop-in for test.py:1065-1126
  dist.py      gallery sharding across GPUs: local top-k + one all-gather + merge
"""
__version__ = "0.1.0"
