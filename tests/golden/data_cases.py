"""Seeded inputs of the data-path golden vectors (shared by make_golden_data.py and the tests)."""
import numpy as np

HIST, QI, E = 6, 2, 32
FIELDS_ALL = ["brand", "category", "main_image", "price", "title"]


def item_samples():
    """14 items with ragged field sets (item 5 has no brand, item 9 only a title); returns (samples, item_dict)."""
    rng = np.random.RandomState(7)
    samples, item_dict = [], {}
    for i in range(14):
        s = {"item_id": f"B{i:03d}", "title": ("Item title %d " % i) + "x" * int(rng.randint(0, 90)), "price": float(rng.rand() * 50)}
        if i != 5:
            s["brand"] = f"brand{i % 3}"
        if i != 9:
            s["category"] = f"cat{i % 4}"
            s["main_image"] = f"http://img/{i}.jpg"
        else:
            s.pop("price")
        samples.append(s)
        item_dict[s["item_id"]] = s
    return samples, item_dict


class FakeItemEncoder:
    """encode_batch_by_field(samples, fields) -> {field: np.float32[B,E]}: a deterministic vector per (item, field),
    all zeros when the sample lacks the field (the reference derives the mask from np.any, qformer_utils.py:116)."""
    embedding_dim = E

    def encode_batch_by_field(self, samples, fields):
        out = {}
        for f in fields:
            rows = []
            for s in samples:
                if f in s and s[f] is not None:
                    seed = (hash_str(s["item_id"]) * 131 + hash_str(f)) % (2 ** 31)
                    v = np.random.RandomState(seed).randn(E).astype(np.float32)
                    rows.append(v / np.linalg.norm(v))
                else:
                    rows.append(np.zeros(E, dtype=np.float32))
            out[f] = np.stack(rows)
        return out


def hash_str(s):
    h = 0
    for c in s:
        h = (h * 31 + ord(c)) % 1000003
    return h


def histories():
    return [["B001", "B005", "B009"], [], ["B013", "B404", "B002", "B003", "B004", "B006", "B007", "B008"], ["B009"] * HIST]


def mrr_inputs():
    rng = np.random.RandomState(11)
    D = 48
    users = rng.randn(5, D).astype(np.float32)
    pos = (users * 0.6 + rng.randn(5, D) * 0.8).astype(np.float32)
    negs = [rng.randn(n, D).astype(np.float32) for n in (7, 1, 19, 12, 3)]
    negs[2][4] = users[2] * 3.0          # a negative that beats the positive
    return users, pos, negs


# ---- event context ----
CTX_H, CTX_SEED = 128, 2024


def context_inputs():
    """Unix timestamps (seconds; python ints) and (lat, lon) degrees."""
    rng = np.random.RandomState(5)
    ts = [int(t) for t in (1.2e9 + rng.rand(9) * 5e8)]
    co = [[float(rng.uniform(-89, 89)), float(rng.uniform(-179, 179))] for _ in range(9)]
    return ts, co


def event_tokens():
    """cached item query tokens item_id -> [QI, CTX_H] f32 (bf16-representable values, so a bf16 cache is exact)."""
    import torch
    rng = np.random.RandomState(9)
    out = {}
    for i in range(14):
        t = torch.from_numpy((rng.randn(QI, CTX_H) * 0.8).astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
        out[f"B{i:03d}"] = t
    return out


def user_events():
    """two users: 5 and 2 events, each {'item_id', 'item_data', 'timestamp', 'coordinates'}."""
    ts, co = context_inputs()
    ids = ["B001", "B005", "B009", "B013", "B002", "B003", "B004"]
    ev = [{"item_id": ids[i], "item_data": {"item_id": ids[i]}, "timestamp": ts[i], "coordinates": co[i]} for i in range(7)]
    return [ev[:5], ev[5:7]]
