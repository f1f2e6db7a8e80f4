"""Prompt construction and ranking evaluation next to the joint head.

  * ``construct_input_text``      training/train_item_individual_token_joint.py:579-592 (history prompt with the
                                  ``<|history_item_i_query_j|>`` specials the model overwrites with Q-Former tokens)
  * ``special_token_positions``   :160-171 (where each special sits in ``input_ids``; the product path finds them inside
                                  ``ur_embed_inject_fwd``, this table is the host-visible form for inspection / tests)
  * ``MRREvaluator``              :361-419 (per-user candidate lists: positive first, cosine scores, rank of the positive)
  * ``CatalogEvaluator``          the same metric with pool = ALL items: one shared catalogue [N,D] resident in HBM,
                                  ``ur_catalog_scores`` + ``ur_rank_of_index`` + ``ur_topk``; no host sync per user.
Tie rule: the positive's rank is 1 + #{strictly greater}; top-K lists the lowest index first.
"""
import numpy as np
import torch

from . import hip

F32 = torch.float32


def construct_input_text(history, item_dict, num_history_items, num_query_tokens_per_item):
    """:579-592 verbatim semantics: numbered titles (truncated to 77 chars + '...') followed by the item's specials;
    empty slots contribute their specials only."""
    history_parts = []
    for i in range(num_history_items):
        query_token_part = "".join(f" <|history_item_{i}_query_{j}|>" for j in range(num_query_tokens_per_item))
        if i < len(history):
            item_id = history[i]
            title = item_dict.get(item_id, {}).get("title", f"Item {item_id}")
            if len(title) > 80:
                title = title[:77] + "..."
            history_parts.append(f"{i + 1}. {title}{query_token_part}")
        else:
            history_parts.append(query_token_part.strip())
    return f"I have bought these items in the past: {', '.join(history_parts)}"


def special_token_positions(input_ids, first_special_id, num_specials):
    """[B, num_specials] int64: position of special id first_special_id + t in each row, -1 when the tokenizer
    truncated it away (:166-171 overwrite only the specials that are present).  Device-side (no per-sample nonzero)."""
    ids = input_ids.to(torch.int64)
    B, S = ids.shape
    rel = ids - int(first_special_id)
    hit = (rel >= 0) & (rel < num_specials)
    pos = torch.full((B, num_specials), -1, dtype=torch.int64, device=ids.device)
    b_idx, s_idx = hit.nonzero(as_tuple=True)
    pos[b_idx, rel[b_idx, s_idx]] = s_idx
    return pos


class MRREvaluator:
    """:361-419.  ``model(input_ids=..., attention_mask=..., history_field_embeddings=..., history_attention_mask=...)``
    -> user embeddings [B,D]; candidates per user: one positive + a ragged list of negatives."""

    def __init__(self, model, tokenizer=None, validation_dataset=None):
        self.model, self.tokenizer, self.validation_dataset = model, tokenizer, validation_dataset

    def _validation_collate_fn(self, batch):
        """:380-395 (negatives stay a list: they are ragged)."""
        st = lambda k: torch.stack([item[k] for item in batch])
        return {"input_ids": st("input_ids"), "attention_mask": st("attention_mask"),
                "history_field_embeddings": st("history_field_embeddings"), "history_attention_mask": st("history_attention_mask"),
                "positive_item_embeddings": st("positive_item_embedding"),
                "negative_item_embeddings": [item["negative_item_embeddings"] for item in batch]}

    @staticmethod
    def ranks_from_embeddings(user_embeddings, positive_item_embeddings, negative_item_embeddings):
        """:403-419 on device: ragged negatives are padded and masked; returns int32 ranks [B]."""
        u = user_embeddings.detach().to(F32).contiguous()
        dev = u.device
        p = torch.as_tensor(positive_item_embeddings).to(dev, F32).contiguous()
        negs = [torch.as_tensor(n).to(dev, F32).reshape(-1, u.shape[1]) for n in negative_item_embeddings]
        nmax = max([n.shape[0] for n in negs] + [1])
        neg = torch.zeros((len(negs), nmax, u.shape[1]), dtype=F32, device=dev)
        mask = torch.zeros((len(negs), nmax), dtype=torch.uint8, device=dev)
        for b, n in enumerate(negs):
            neg[b, :n.shape[0]] = n
            mask[b, :n.shape[0]] = 1
        scores, _ = hip.cosine_scores(u, p, neg)
        return hip.mrr_rank(scores, mask)

    def _compute_batch_mrr(self, batch):
        dev = next(self.model.parameters()).device
        user = self.model(input_ids=batch["input_ids"].to(dev), attention_mask=batch["attention_mask"].to(dev),
                          history_field_embeddings=batch["history_field_embeddings"].to(dev),
                          history_attention_mask=batch["history_attention_mask"].to(dev))
        rank = self.ranks_from_embeddings(user, batch["positive_item_embeddings"], batch["negative_item_embeddings"])
        return (1.0 / rank.to(torch.float64)).cpu().tolist()

    def evaluate_mrr(self, batch_size: int = 32) -> float:
        """:367-378."""
        self.model.eval()
        scores = []
        loader = torch.utils.data.DataLoader(self.validation_dataset, batch_size=batch_size, shuffle=False,
                                             collate_fn=self._validation_collate_fn)
        with torch.no_grad():
            for batch in loader:
                scores.extend(self._compute_batch_mrr(batch))
        return float(np.mean(scores))


class CatalogEvaluator:
    """MRR / hit@K / top-K of user embeddings against the whole item catalogue (pool = all items)."""

    def __init__(self, catalog, item_ids=None, device="cuda"):
        self.catalog = torch.as_tensor(catalog).to(device, F32).contiguous()      # [N,D], stays in HBM
        self.item_ids = None if item_ids is None else [str(i) for i in item_ids]
        self._inv = None                                                          # catalogue 1/||c||, computed once

    def scores(self, user_embeddings):
        s, self._inv = hip.catalog_scores(user_embeddings.detach().to(self.catalog.device, F32).contiguous(), self.catalog, self._inv)
        return s

    def evaluate(self, user_embeddings, gt_index, k=10):
        """-> dict(rank int32 [B], mrr float, hit_at_k float, topk_index int32 [B,k], topk_score f32 [B,k])."""
        s = self.scores(user_embeddings)
        gt = torch.as_tensor(gt_index).to(s.device, torch.int64)
        rank = hip.rank_of_index(s, gt)
        idx, val = hip.topk(s, k)
        return {"rank": rank, "mrr": float((1.0 / rank.to(torch.float64)).mean().item()),
                "hit_at_k": float((rank <= k).to(torch.float64).mean().item()), "topk_index": idx, "topk_score": val}
