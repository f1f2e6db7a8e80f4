"""CPU restatement (TEST INFRASTRUCTURE, never imported by unirec_amd/) of the reference's data path either side of
the hot path -- SURVEY.md section 8(f) rows N1 / N2 / N4.  Plain Python / numpy loops, each function citing the
reference lines it follows.  Pinned against the reference itself by tests/golden/data_path.npz + data_path.json
(tests/golden/make_golden_data.py runs the reference's classes in the build container)."""
import numpy as np


def analyze_fields(samples):
    """models/qformer_utils.py:83-96."""
    all_fields = set()
    for s in samples:
        all_fields.update(s.keys())
    return sorted(f for f in all_fields if f != "item_id")


def precompute_cache(samples, encode_batch_by_field, fields, batch_size):
    """models/qformer_utils.py:98-124 -> (embeddings [N,F,E] f32, masks [N,F] i64): valid iff the vector has a non-zero."""
    embs, masks = [], []
    for i in range(0, len(samples), batch_size):
        batch = samples[i:i + batch_size]
        enc = encode_batch_by_field(batch, fields)
        for j in range(len(batch)):
            e = [np.asarray(enc[f][j], dtype=np.float32) for f in fields]
            embs.append(np.stack(e))
            masks.append(np.array([1 if np.any(v) else 0 for v in e], dtype=np.int64))
    return np.stack(embs), np.stack(masks)


def history_qformer_inputs(history_item_ids, item_id_to_idx, embeddings, masks, num_history_items):
    """training/train_item_individual_token_joint.py:557-577 for ONE sample -> ([hist,F,E] f32, [hist,F] i64)."""
    F, E = embeddings.shape[1], embeddings.shape[2]
    out_e = np.zeros((num_history_items, F, E), dtype=np.float32)
    out_m = np.zeros((num_history_items, F), dtype=np.int64)
    for i in range(num_history_items):
        if i < len(history_item_ids):
            item_id = str(history_item_ids[i])
            if item_id in item_id_to_idx:
                out_e[i] = embeddings[item_id_to_idx[item_id]]
                out_m[i] = masks[item_id_to_idx[item_id]]
    return out_e, out_m


def construct_input_text(history, item_dict, num_history_items, num_query_tokens_per_item):
    """training/train_item_individual_token_joint.py:579-592."""
    parts = []
    for i in range(num_history_items):
        q = "".join([f" <|history_item_{i}_query_{j}|>" for j in range(num_query_tokens_per_item)])
        if i < len(history):
            item_id = history[i]
            title = item_dict.get(item_id, {}).get("title", f"Item {item_id}")
            if len(title) > 80:
                title = title[:77] + "..."
            parts.append(f"{i + 1}. {title}{q}")
        else:
            parts.append(q.strip())
    return f"I have bought these items in the past: {', '.join(parts)}"


def history_query_tokens(history_items, token_dict, num_history_items, num_query_tokens_per_item, dim):
    """training/train_item_individual_token_joint.py:241-255 -> [hist,Q,D] f32 (zeros for empty slots / unknown items)."""
    out = np.zeros((num_history_items, num_query_tokens_per_item, dim), dtype=np.float32)
    for i, item_id in enumerate(history_items[:num_history_items]):
        out[i] = token_dict.get(item_id, np.zeros((num_query_tokens_per_item, dim)))
    return out


def _normalize(x, eps=1e-12):
    """torch.nn.functional.normalize(p=2, dim=-1)."""
    n = np.sqrt((x.astype(np.float32) ** 2).sum(-1, keepdims=True, dtype=np.float32))
    return (x / np.maximum(n, eps)).astype(np.float32)


def mrr_rank(user, pos, negs):
    """training/train_item_individual_token_joint.py:403-419 for one user: 1-based rank of the positive among
    [positive; negatives] by cosine similarity.  Ties: 1 + #{strictly greater} (the reference's argsort is unstable)."""
    u, p, n = _normalize(user), _normalize(pos), _normalize(negs)
    sims = np.concatenate([p[None], n], 0) @ u
    return 1 + int((sims[1:] > sims[0]).sum()), sims


def catalog_eval(users, catalog, gt_index, k):
    """pool = all items: cosine scores [B,N], rank of column gt (ties to the ground truth), top-k (lowest index first)."""
    s = _normalize(users) @ _normalize(catalog).T
    rank = np.array([1 + int((s[b] > s[b, gt_index[b]]).sum()) for b in range(s.shape[0])], dtype=np.int64)
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    return s, rank, order


# ---- event context encoders ---------------------------------------------------------------------------
def context_mlp_shapes(H, nfeat):
    return {"projection.0.weight": (2 * H, nfeat), "projection.0.bias": (2 * H,), "projection.2.weight": (H, 2 * H), "projection.2.bias": (H,)}


def timestamp_features(ts):
    """models/mwne.py:525-565, every step in float32 exactly as torch evaluates it on ``timestamps.float()``."""
    f32 = np.float32
    x = np.asarray(ts, dtype=np.float64).astype(f32).reshape(-1, 1)
    year, day, two_pi = f32(365.25 * 24 * 60 * 60), f32(24 * 60 * 60), f32(2 * np.pi)
    comps = [x / year]
    day_phase = np.mod(x, day) / day
    comps += [np.sin(two_pi * day_phase), np.cos(two_pi * day_phase)]
    week_phase = ((x / day) + f32(4)) / f32(7)
    comps += [np.sin(two_pi * week_phase), np.cos(two_pi * week_phase)]
    year_phase = np.mod(x, year) / year
    comps += [np.sin(two_pi * year_phase), np.cos(two_pi * year_phase)]
    month_phase = year_phase * f32(12)
    comps += [np.sin(two_pi * month_phase), np.cos(two_pi * month_phase)]
    return np.concatenate([c.astype(f32) for c in comps], axis=-1)


def geo_features(coords):
    """models/mwne.py:586-607."""
    c = np.asarray(coords, dtype=np.float32)
    lat, lon = np.deg2rad(c[:, 0]).astype(np.float32), np.deg2rad(c[:, 1]).astype(np.float32)
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=-1).astype(np.float32)


def context_mlp(feat, P):
    """nn.Sequential(Linear, GELU(erf), Linear) in f32 (models/mwne.py:519-523, :579-583)."""
    from math import erf
    h = feat @ P["projection.0.weight"].T + P["projection.0.bias"]
    g = 0.5 * h * (1.0 + np.vectorize(erf)(h / np.sqrt(2.0)))
    return (g.astype(np.float32) @ P["projection.2.weight"].T + P["projection.2.bias"]).astype(np.float32)
