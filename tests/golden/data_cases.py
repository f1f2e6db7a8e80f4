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
