"""Flat parameter storage for the live (gradient-receiving) tensors of a module.

All live parameters of a model are views into ONE fp32 master buffer; a bf16 shadow (the GEMM
operands) and an fp32 gradient buffer have the same element layout.  Consequences, all of them the
point of the design:
  * fp32->bf16 refresh is one kernel per step, AdamW is one kernel per step,
  * the data-parallel gradient all-reduce runs over a few large contiguous buckets,
  * adjacent tensors (query|key|value weights, key|value weights, gate|up ...) are one fused GEMM
    operand without any copy,
  * the reference's dead tensors (word/position embeddings, text FFN: SURVEY §8(a) I1) are simply not
    in the pack: they never cost optimizer or all-reduce bytes, but keep their state_dict keys.
PyTorch is only the allocator here; the arithmetic is in libunirec_hip.so.
"""
import torch

from . import hip

ALIGN = 8   # elements: keeps every bf16 view 16-byte aligned


def norm_device(device):
    """torch.device with an explicit index ("cuda" -> the current device): packs compare devices exactly, and a forward sees
    its inputs' indexed device, so an index-less handle must not look like another device (it would rebuild the pack)."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


class ParamPack:
    def __init__(self, named_params, device):
        """named_params: ordered list of (name, nn.Parameter).  Adjacent entries are adjacent in memory
        whenever their sizes are multiples of ALIGN."""
        self.names = [n for n, _ in named_params]
        self.params = {n: p for n, p in named_params}
        self.offsets, self.shapes = {}, {}
        off = 0
        for n, p in named_params:
            self.offsets[n] = off
            self.shapes[n] = tuple(p.shape)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self._views = {}
        self.device = norm_device(device)
        self.master = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        for n, p in named_params:
            v = self._view(self.master, n)
            v.copy_(p.data.to(device=self.device, dtype=torch.float32))
            p.data = v
        self._shadow_key = None
        self.dirty = True
        self.live = set()          # names whose gradient entry a backward has written since the last clear_grads()

    def _view(self, flat, name, shape=None):
        # (views are cached: a Q-Former step asks for ~330 of them, 4 us of slicing + reshaping each; the flat buffers never move)
        key = (id(flat), name, shape)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._make_view(flat, name, shape)
        return v

    def _make_view(self, flat, name, shape=None):
        o = self.offsets[name]
        shp = self.shapes[name] if shape is None else shape
        n = 1
        for s in shp:
            n *= s
        return flat[o:o + n].view(shp)

    def is_current(self):
        """False when .to()/load of a different storage re-pointed the parameters away from the pack."""
        n0 = self.names[0]
        p = self.params[n0]
        return p.data.data_ptr() == self.master.data_ptr() + 4 * self.offsets[n0] and p.device == self.device

    def w16(self, name):
        return self._view(self.shadow, name)

    def w32(self, name):
        return self._view(self.master, name)

    def g32(self, name):
        return self._view(self.grad, name)

    def fused(self, flat, names, rows_each=None):
        """One 2-D (or 1-D) view spanning several ADJACENT tensors with equal trailing shape."""
        key = (id(flat), tuple(names))
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._make_fused(flat, names)
        return v

    def _make_fused(self, flat, names):
        first = names[0]
        o = self.offsets[first]
        total_rows = 0
        trailing = self.shapes[first][1:]
        exp = o
        for n in names:
            if self.offsets[n] != exp or self.shapes[n][1:] != trailing:
                raise RuntimeError(f"parameters {names} are not contiguous in the pack")
            cnt = 1
            for s in self.shapes[n]:
                cnt *= s
            if cnt % ALIGN:
                raise RuntimeError(f"{n}: size {cnt} breaks contiguity (not a multiple of {ALIGN})")
            exp += cnt
            total_rows += self.shapes[n][0]
        shape = (total_rows,) + trailing
        return flat[o:exp].view(shape)

    def fused16(self, names):
        return self.fused(self.shadow, names)

    def fused32(self, names):
        return self.fused(self.master, names)

    def fusedg(self, names):
        return self.fused(self.grad, names)

    def mark_dirty(self):
        self.dirty = True

    def refresh_shadow(self, force=False):
        """fp32 master -> bf16 shadow when any parameter changed (torch's in-place version counters
        catch torch optimizers / load_state_dict; mark_dirty() is for the fused HIP optimizer)."""
        key = sum(p._version for p in self.params.values())
        if force or self.dirty or key != self._shadow_key:
            hip.cast_f32_to_bf16(self.master, self.shadow)
            self._shadow_key = key
            self.dirty = False

    def publish_grads(self, names=None):
        """Expose freshly written entries of the flat gradient buffer through param.grad (torch.optim /
        user code see ordinary per-tensor gradients; parameters a backward did not touch keep
        grad=None exactly as in the reference).  Gradients are OVERWRITTEN per backward (the reference
        always zero_grads first: training/item_qformer_training.py:129); a param holding a foreign
        gradient tensor gets the new value accumulated (cold path)."""
        names = self.names if names is None else names
        self.live.update(names)
        for n in names:
            p = self.params[n]
            g = self.g32(n)
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)

    def clear_grads(self, set_to_none=True):
        """optimizer.zero_grad(): forget which entries are live; param.grad -> None (or zeroed views)."""
        for n in self.live:
            p = self.params[n]
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        self.live = set()

    def span(self, name):
        """[lo, hi) of one tensor in the flat buffers, padding up to the next tensor included."""
        lo = self.offsets[name]
        cnt = 1
        for d in self.shapes[name]:
            cnt *= d
        return lo, lo + (cnt + ALIGN - 1) // ALIGN * ALIGN

    def live_ranges(self, key=None):
        """Maximal contiguous [lo, hi) runs of live tensors (in pack order); `key(name)` splits runs whose tensors must
        not share a launch (different optimizer step counts)."""
        runs = []
        for n in self.names:
            if n not in self.live:
                continue
            lo, hi = self.span(n)
            k = None if key is None else key(n)
            if runs and runs[-1][1] == lo and runs[-1][2] == k:
                runs[-1][1] = hi
            else:
                runs.append([lo, hi, k])
        return [(lo, hi, k) for lo, hi, k in runs]
