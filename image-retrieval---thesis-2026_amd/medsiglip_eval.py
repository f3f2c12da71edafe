"""Zero-shot classification + all-pairs retrieval evaluation of a SigLIP-style dual encoder.

Mirrors (paths into /root/reference):
  COVIDX_LABEL_TO_TEXT                               train_medsiglip.py:21-25
  get_text_features(model, processor, device, prompts, max_text_length)   eval_medsiglip.py:163-186
  evaluate(model, processor, loader, device, args)                        eval_medsiglip.py:189-260
Same stages and prints as the reference: class prompts -> unit-norm text features; per batch image features -> unit
norm -> logits = exp(logit_scale) * img @ txt.T -> argmax; then accuracy / macro P / R / F1 of the zero-shot
predictions and the retrieval tail (R@K, mAP, mP@K, majority-vote classification) over the cosine similarities of the
image features with the diagonal excluded.  What changes underneath: the image features never leave the GPU, the
N x N similarity matrix, its topk and its argsort are replaced by one resident FlatIndex and libmirx's exact full
ranking (fp64 scores, ties -> lowest id), and AP / precision@k run over that ranking on the device.  `model` is anything
with the transformers SiglipModel surface (mirx.siglip.SiglipDualEncoder, or the reference's own object); `processor`
anything whose `.tokenizer(prompts, max_length=..., padding="max_length", truncation=True, return_attention_mask=True,
return_tensors="pt")` returns input_ids / attention_mask.  Returns the numbers it prints (the reference returns None).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .evaluate import rank_self
from .metrics import _prf, compute_classification_metrics, compute_map, retrieval_accuracy

COVIDX_LABEL_TO_TEXT = {
    0: "A chest X-ray showing no evidence of pneumonia or COVID-19 infection.",
    1: "A chest X-ray showing findings consistent with pneumonia.",
    2: "A chest X-ray showing findings consistent with COVID-19 pneumonia.",
}


def _encode(model, kind, **inputs):
    """Unit-norm features of one tower.  kind = "text" | "image"; `inputs` are that tower's keyword tensors.  Models with the
    SiglipModel surface answer through get_<kind>_features (a tensor, or -- transformers >= 5 -- an output object whose
    <kind>_embeds / pooler_output holds it); anything else is called whole and its <kind>_embeds taken (eval_medsiglip.py:174-184,
    205-211 accept both)."""
    getter = getattr(model, f"get_{kind}_features", None)
    out = getter(**inputs) if getter is not None else model(**inputs)
    if not torch.is_tensor(out):
        emb = getattr(out, f"{kind}_embeds", None)
        out = emb if emb is not None else out.pooler_output
    return F.normalize(out, dim=-1)


@torch.no_grad()
def get_text_features(model, processor, device, prompts, max_text_length):
    """eval_medsiglip.py:163-186: tokenise to a fixed length, encode, unit-normalise."""
    tok = processor.tokenizer(prompts, max_length=max_text_length, padding="max_length", truncation=True,
                              return_attention_mask=True, return_tensors="pt").to(device)
    return _encode(model, "text", input_ids=tok["input_ids"], attention_mask=tok["attention_mask"])


@torch.no_grad()
def evaluate(model, processor, loader, device, args, label_to_text=None):
    model.eval()
    label_to_text = label_to_text or COVIDX_LABEL_TO_TEXT
    text_features = get_text_features(model, processor, device, [label_to_text[i] for i in sorted(label_to_text)],
                                      args.max_text_length)

    # image tower over the loader: features and labels stay on the device; the zero-shot decision is ONE product over all of
    # them afterwards (the reference decides batch by batch and ships every batch to the host: eval_medsiglip.py:201-219)
    feats, labs = [], []
    for done, batch in enumerate(loader, start=1):
        feats.append(_encode(model, "image", pixel_values=batch["pixel_values"].to(device)))
        labs.append(batch["labels"].to(device))
        if done % 10 == 0:
            print(f"Processed {done * args.eval_batch_size} images...")
    embeds = torch.cat(feats, dim=0).float()
    labels = torch.cat(labs).long()
    scale = model.logit_scale.exp() if hasattr(model, "logit_scale") else torch.tensor(100.0, device=device)
    all_predictions = torch.argmax(scale * embeds @ text_features.float().t(), dim=-1).cpu().numpy()
    all_labels = labels.cpu().numpy()

    p, r, f, _ = _prf(all_labels, all_predictions)          # sklearn macro averages, zero_division=0
    zs = {"accuracy": float(np.mean(all_predictions == all_labels) * 100.0), "precision_macro": float(p.mean() * 100.0),
          "recall_macro": float(r.mean() * 100.0), "f1_macro": float(f.mean() * 100.0)}
    print("\n>> Zero-shot Classification Metrics:")
    print(f"   Accuracy: {zs['accuracy']:.2f}%")
    print(f"   Precision (macro): {zs['precision_macro']:.2f}%")
    print(f"   Recall (macro): {zs['recall_macro']:.2f}%")
    print(f"   F1 (macro): {zs['f1_macro']:.2f}%")

    # dists = embeds @ embeds.t(), diagonal -inf (eval_medsiglip.py:238-239): one resident index + exact full ranking
    ranks, _ = rank_self(embeds, "cosine")
    k_values = [1, 5, 10, 15, 20]
    head = ranks[:, :max(k_values)].cpu().numpy()
    kappas = [1, 5, 10]
    accuracy = torch.stack(retrieval_accuracy(None, all_labels, topk=kappas, topk_ids=head[:, :max(kappas)])).cpu().numpy()
    print(f">> R@K{kappas}: {np.around(accuracy, 2)}%")
    m_ap, aps, pr, _ = compute_map(ranks.t(), all_labels, kappas)
    print(f">> mAP: {m_ap * 100.0:.2f}%")
    print(f">> mP@K{kappas}: {np.around(pr * 100.0, 2)}%")
    print("\n>> Retrieval Classification Metrics (Majority Voting):")
    classification_results = compute_classification_metrics(all_labels, None, k_values, ranks=head.T)
    for k in k_values:
        m = classification_results[k]
        print(f"\n>> Top-{k} Retrieved Images:")
        print(f"   Accuracy: {m['accuracy']:.2f}%")
        print(f"   Precision (macro): {m['precision_macro']:.2f}%")
        print(f"   Recall (macro): {m['recall_macro']:.2f}%")
        print(f"   F1 (macro): {m['f1_macro']:.2f}%")
    return {"zero_shot": zs, "predictions": all_predictions, "labels": all_labels, "text_features": text_features,
            "embeds": embeds, "ranks": ranks, "acc": accuracy, "mAP": m_ap, "aps": aps, "pr": pr,
            "classification": classification_results}
