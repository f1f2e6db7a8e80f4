"""Drop-in for the model/loss part of ``training/train_item_individual_token_joint.py``:
``MultiModalQwenEmbedding`` (:88-212), ``InfoNCELoss`` (:326-352), MRR (:392-419) and
``MultiModalTrainer.compute_loss`` (:482-498), all on the HIP path.

Differences that are deliberate and documented:
  * no network here: the Qwen3 backbone is built from a ``Qwen3Config`` (random init or weights the
    caller loads with load_state_dict) instead of ``AutoModel.from_pretrained``; ``.tokenizer`` is a
    minimal table of the history special tokens unless a real tokenizer is handed in.
  * ``num_history_items`` / ``num_query_tokens_per_item`` are constructor arguments (the reference
    hard-codes 10 and 2, :93-94, which cannot express BASELINE's hist=50/100).
  * the python triple loop with one host sync per (item, query, sample) (:160-171) is one kernel.
"""
import json
import os

import torch
import torch.nn as nn

from . import hip
from .qwen3 import Qwen3Config, Qwen3LoRAModel

BF16, F32 = torch.bfloat16, torch.float32


class HistoryTokenTable:
    """Stand-in for the tokenizer methods the joint model touches (:106-116,163)."""

    def __init__(self, base_vocab, num_history_items, num_query_tokens_per_item):
        self.base_vocab = base_vocab
        self.history_tokens = [f"<|history_item_{i}_query_{j}|>" for i in range(num_history_items)
                               for j in range(num_query_tokens_per_item)]
        self._ids = {t: base_vocab + k for k, t in enumerate(self.history_tokens)}
        self.pad_token = "<|endoftext|>"

    def __len__(self):
        return self.base_vocab + len(self.history_tokens)

    def convert_tokens_to_ids(self, name):
        return self._ids[name]

    def save_pretrained(self, save_directory):
        with open(os.path.join(save_directory, "history_tokens.json"), "w") as f:
            json.dump(self._ids, f)


class MultiModalQwenEmbedding(nn.Module):
    def __init__(self, base_model_name: str = "Qwen/Qwen3-Embedding-0.6B", qformer_model: nn.Module = None, use_lora: bool = True,
                 lora_config=None, qwen_config: Qwen3Config = None, num_history_items: int = 10,
                 num_query_tokens_per_item: int = 2, tokenizer=None, user_qformer: nn.Module = None):
        super().__init__()
        # U4 (README.md:44-47 / figure (c), no reference code): the User Q-Former's query
        # tokens are injected at <|user_query_k|> specials that follow the history specials.
        self.user_qformer = user_qformer
        self.num_user_query_tokens = 0 if user_qformer is None else int(user_qformer.num_query_tokens)
        self.use_lora = use_lora
        self.num_history_items = num_history_items
        self.num_query_tokens_per_item = num_query_tokens_per_item
        self.qformer_model = qformer_model
        cfg = qwen_config or Qwen3Config()
        if lora_config is not None:      # peft.LoraConfig-like object or dict: r / lora_alpha / lora_dropout
            get = (lambda k, d: lora_config.get(k, d)) if isinstance(lora_config, dict) else (lambda k, d: getattr(lora_config, k, d))
            cfg.lora_r, cfg.lora_alpha, cfg.lora_dropout = get("r", cfg.lora_r), get("lora_alpha", cfg.lora_alpha), get("lora_dropout", cfg.lora_dropout)
        self.base_model = Qwen3LoRAModel(cfg, use_lora=use_lora)
        self.hidden_size = cfg.hidden_size
        if qformer_model is not None and qformer_model.config.hidden_size != self.hidden_size:
            raise ValueError("No projector: Q-Former hidden size must equal the LLM hidden size (:109)")
        base_vocab = cfg.vocab_size
        self.history_tokens = [f"<|history_item_{i}_query_{j}|>" for i in range(num_history_items)
                               for j in range(num_query_tokens_per_item)]
        self.user_tokens = [f"<|user_query_{k}|>" for k in range(self.num_user_query_tokens)]
        if tokenizer is None:
            tokenizer = HistoryTokenTable(base_vocab, num_history_items, num_query_tokens_per_item)
        elif hasattr(tokenizer, "add_special_tokens"):
            # a real tokenizer: extend it exactly as the reference does (:106-118); ids of the added tokens are consecutive
            if getattr(tokenizer, "pad_token", None) is None and getattr(tokenizer, "eos_token", None) is not None:
                tokenizer.pad_token = tokenizer.eos_token
            tokenizer.add_special_tokens({"additional_special_tokens": self.history_tokens + self.user_tokens})
        self.tokenizer = tokenizer
        # ids of the added special tokens are consecutive (tokenizer.add_special_tokens order, :106-119)
        self.first_special_id = int(self.tokenizer.convert_tokens_to_ids(self.history_tokens[0]))
        self.first_user_special_id = self.first_special_id + len(self.history_tokens)
        need = self.first_user_special_id + len(self.user_tokens)
        try:
            need = max(need, len(self.tokenizer))          # :119 resize_token_embeddings(len(tokenizer))
        except TypeError:
            pass
        self.base_model.resize_token_embeddings(max(base_vocab, need))
        if user_qformer is not None and user_qformer.config.hidden_size != self.hidden_size:
            raise ValueError("No projector: User Q-Former hidden size must equal the LLM hidden size")

    def forward(self, input_ids, attention_mask=None, history_field_embeddings=None, history_attention_mask=None,
                user_sequence_tokens=None, user_attention_mask=None):
        dev = self.base_model.embed_tokens.weight.device
        input_ids = input_ids.to(dev)
        if attention_mask is not None:
            attention_mask = attention_mask.to(dev)
        item_tokens = None
        if self.use_lora and self.training and input_ids.is_cuda:
            # this step's LoRA dropout bit planes do not depend on any activation: generate them on a side stream under the
            # Q-Former forward(s) below
            self.base_model.prefetch_lora_bits(input_ids.shape[0] * input_ids.shape[1], dev,
                                               row0=self.base_model.first_sample(input_ids.shape[0]) * input_ids.shape[1])
        if history_field_embeddings is not None and history_attention_mask is not None:
            hfe = history_field_embeddings.to(dev)
            ham = history_attention_mask.to(dev)
            bh, num_hist, num_fields, field_dim = hfe.shape
            h16 = self.qformer_model.encode_bf16(hfe.reshape(bh * num_hist, num_fields, field_dim), ham.reshape(bh * num_hist, num_fields))
            if h16.shape[1] != self.num_query_tokens_per_item or num_hist != self.num_history_items:
                raise ValueError("history layout does not match num_history_items x num_query_tokens_per_item")
            item_tokens = h16.reshape(bh, num_hist * self.num_query_tokens_per_item, self.hidden_size)
        if self.user_qformer is not None and user_sequence_tokens is not None:
            u16 = self.user_qformer.encode_bf16(user_sequence_tokens.to(dev), user_attention_mask.to(dev))     # [B,64,D]
            if item_tokens is None:
                item_tokens = torch.zeros((u16.shape[0], len(self.history_tokens), self.hidden_size), dtype=BF16, device=dev)
            item_tokens = torch.cat([item_tokens, u16], dim=1)      # special ids are consecutive: history block, then user block
        return self.base_model.forward_pooled(input_ids, attention_mask, item_tokens, self.first_special_id)

    def load_base_weights(self, state_dict, strict=True):
        """Load a Qwen3 checkpoint (the state_dict of the ``AutoModel`` the reference builds at :98-103) into the frozen
        backbone AFTER the vocabulary was extended: ``embed_tokens.weight`` of the checkpoint has the base vocabulary only,
        its rows are copied and the added special-token rows are kept (load first, resize afterwards in the reference,
        :99-119 -- same result).  Returns (missing, unexpected) like ``load_state_dict``."""
        return self.base_model.load_base_weights(state_dict, strict=strict)

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        self.tokenizer.save_pretrained(save_directory)
        if self.use_lora:
            torch.save(self.base_model.peft_state_dict(), os.path.join(save_directory, "adapter_model.bin"))
        else:
            torch.save(self.base_model.state_dict(), os.path.join(save_directory, "base_model.bin"))
        torch.save(self.qformer_model.state_dict(), os.path.join(save_directory, "qformer_model.bin"))
        with open(os.path.join(save_directory, "model_config.json"), "w") as f:
            json.dump({"hidden_size": self.hidden_size, "use_lora": self.use_lora}, f, indent=2)
        print(f"Model saved to {save_directory}")

    def get_trainable_parameters(self):
        trainable_params = sum(p.numel() for p in self.parameters() if p.requires_grad)
        all_param = sum(p.numel() for p in self.parameters())
        print(f"Trainable params: {trainable_params:,} || All params: {all_param:,} || "
              f"Trainable%: {100 * trainable_params / all_param:.2f}%")
        return trainable_params, all_param


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user, pos, neg, neg_mask_u8, temperature):
        scores, inv = hip.cosine_scores(user, pos, neg)
        loss, du = hip.infonce_fwd_bwd(user, pos, neg, neg_mask_u8, scores, inv, temperature, 1.0, need_grad=True)
        ctx.du = du
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        du = ctx.du * g     # scalar upstream gradient (1.0 from loss.backward()); elementwise scale of [B,D]
        return du, None, None, None, None


class InfoNCELoss(nn.Module):
    def __init__(self, temperature: float = 0.07):
        super().__init__()
        self.temperature = temperature

    def forward(self, user_embeddings, positive_item_embeddings, negative_item_embeddings, negative_masks=None):
        u = user_embeddings.contiguous().to(F32)
        dev = u.device
        p = positive_item_embeddings.to(dev, F32).contiguous()
        n = negative_item_embeddings.to(dev, F32).contiguous()
        m = None if negative_masks is None else negative_masks.to(dev).to(torch.uint8).contiguous()
        return _InfoNCEFn.apply(u, p, n, m, float(self.temperature))


def mrr_ranks(user_embeddings, positive_item_embeddings, negative_item_embeddings, negative_masks=None):
    """(:408-419) cosine scores of [positive; negatives] and the positive's 1-based rank per user."""
    u = user_embeddings.detach().contiguous().to(F32)
    p = positive_item_embeddings.to(u.device, F32).contiguous()
    n = negative_item_embeddings.to(u.device, F32).contiguous()
    m = None if negative_masks is None else negative_masks.to(u.device).to(torch.uint8).contiguous()
    scores, _ = hip.cosine_scores(u, p, n)
    return scores, hip.mrr_rank(scores, m)


class MultiModalTrainer:
    """compute_loss of the reference's HF-Trainer subclass (:477-498) without the Trainer."""

    def __init__(self, temperature: float = 0.07):
        self.infonce_loss = InfoNCELoss(temperature)

    def compute_loss(self, model, inputs, return_outputs=False, **kwargs):
        user_embeddings = model(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                                history_field_embeddings=inputs["history_field_embeddings"],
                                history_attention_mask=inputs["history_attention_mask"])
        loss = self.infonce_loss(user_embeddings, inputs["positive_item_embeddings"], inputs["negative_item_embeddings"],
                                 inputs.get("negative_masks", None))
        return (loss, user_embeddings) if return_outputs else loss
