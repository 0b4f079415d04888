"""Flat parameter / gradient / Adam-state storage for a module.

All parameters of a net live in ONE fp32 buffer (each tensor 16-byte aligned inside it) and every
nn.Parameter is re-pointed to a view of it; gradients live in a second flat buffer of the same layout.
That gives: one fused Adam launch per step, one RCCL all-reduce per step (reference has neither:
Processor/Train/Train_Upper.py:60,181-182 steps 54 tensors one by one), and float4-aligned weights for
the GEMM kernels.  state_dict keys/shapes are untouched, so shipped checkpoints load unchanged.
"""
import torch

from . import hip

ALIGN = 4  # floats


class FlatParams:
    def __init__(self, module):
        self.module = module
        self.params = [p for p in module.parameters()]
        # A module may ask for a layout of its own (``flat_param_order() -> list of its parameters``): tensors that one product
        # wants to treat as ONE matrix (the weights of two convolutions over the same input, stacked) then sit back to back in
        # the parameter AND gradient buffers.  state_dict keys and shapes are untouched; only the flat offsets move.
        order = getattr(module, "flat_param_order", None)
        if callable(order):
            wanted = list(order())
            if sorted(id(p) for p in wanted) != sorted(id(p) for p in self.params):
                raise ValueError("flat_param_order() must return every parameter of the module exactly once")
            self.params = wanted
        self.device = None
        self.flat_p = self.flat_g = None
        self.offsets = []
        self._gviews = {}
        self._counters = None

    def _layout(self):
        off = 0
        self.offsets = []
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        return off

    def ensure(self):
        """(Re)build the flat buffers if the module moved device or was re-materialised."""
        p0 = self.params[0]
        if (self.flat_p is not None and p0.device == self.device and
                p0.data_ptr() == self.flat_p.data_ptr() + 4 * self.offsets[0] and
                self.params[-1].data_ptr() == self.flat_p.data_ptr() + 4 * self.offsets[-1]):
            return self
        self.device = p0.device
        total = self._layout()
        flat_p = torch.zeros(total, dtype=torch.float32, device=self.device)
        flat_g = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._gviews = {}
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + n].view(p.shape)
            self._gviews[id(p)] = flat_g[off:off + n].view(p.shape)
        self.flat_p, self.flat_g = flat_p, flat_g
        # BatchNorm step counters: one int64 buffer, bumped with a single launch per training forward
        named = [(n, b) for n, b in self.module.named_buffers() if n.endswith("num_batches_tracked")]
        if named:
            flat_c = torch.stack([b.to(self.device).reshape(()) for _, b in named]).contiguous()
            for i, (name, _) in enumerate(named):
                owner_name, _, leaf = name.rpartition(".")
                self.module.get_submodule(owner_name)._buffers[leaf] = flat_c[i]
            self._counters = flat_c
        return self

    def grad(self, p):
        return self._gviews[id(p)]

    def rebind_grad(self, buf):
        """Move the flat gradient buffer into ``buf`` (1-D fp32, same length, 16-byte aligned): several nets' gradients can
        then live back to back in ONE buffer and be summed over the ranks by ONE collective (train_step.GradBucket).  Must
        happen before anything captures the old addresses (HIP graphs)."""
        self.ensure()
        if buf.numel() != self.flat_g.numel() or buf.dtype != torch.float32 or buf.device != self.flat_g.device or buf.data_ptr() % 16:
            raise ValueError("rebind_grad: need an aligned fp32 buffer of %d elements on %s" % (self.flat_g.numel(), self.flat_g.device))
        buf.copy_(self.flat_g)
        self.flat_g = buf
        for p, off in zip(self.params, self.offsets):
            self._gviews[id(p)] = buf[off:off + p.numel()].view(p.shape)
            if p.grad is not None:
                p.grad = self._gviews[id(p)]

    def tick_args(self, seed_ctr=None):
        """(counters, n, seed_ctr): the once-per-training-forward tick of a net -- BatchNorm num_batches_tracked += 1 for all
        layers and the next dropout seed -- as the arguments of the kernel that carries it (mmego_head_fk_forward; mmego_inc_i64
        is the same tick as a launch of its own)."""
        if self._counters is not None:
            return (self._counters, self._counters.numel(), seed_ctr)
        return (None, 0, seed_ctr)

    def bind_grads(self):
        """Expose the flat gradient views as ``p.grad`` (for torch optimisers / inspection)."""
        for p in self.params:
            if p.requires_grad:
                p.grad = self._gviews[id(p)]


class FusedAdam:
    """torch.optim.Adam semantics (betas 0.9/0.999, eps 1e-8, coupled weight decay) in one launch."""

    def __init__(self, flat, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = self.v = self.state = self._ticket = None
        self._skip = None

    def _skip_ranges(self, f):
        """Element ranges of parameters that never receive a gradient (`module.never_trained()`, e.g. IMU_Net.fc3):
        torch.optim.Adam leaves a parameter with grad=None alone -- no moment update, no weight decay."""
        if self._skip is None:
            dead = {id(p) for p in getattr(f.module, "never_trained", lambda: ())()}
            spans = []
            for p, off in zip(f.params, f.offsets):
                if id(p) in dead:
                    end = off + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                    if spans and spans[-1][1] == off:
                        spans[-1][1] = end
                    else:
                        spans.append([off, end])
            if len(spans) > 4:
                raise ValueError("FusedAdam supports at most 4 disjoint never-trained parameter ranges")
            self._skip = torch.tensor([v for sp in spans for v in sp], dtype=torch.int64) if spans else False
        return self._skip

    def _ensure(self):
        f = self.flat.ensure()
        if self.m is None or self.m.device != f.device or self.m.numel() != f.flat_p.numel():
            self.m = torch.zeros_like(f.flat_p)
            self.v = torch.zeros_like(f.flat_p)
            self.state = torch.zeros(3, dtype=torch.float64, device=f.device)
            self._ticket = torch.zeros(1, dtype=torch.int32, device=f.device)
        return f

    def step(self):
        f = self._ensure()
        skip = self._skip_ranges(f)
        hip.call("adam_step", f.flat_p, f.flat_g, self.m, self.v, f.flat_p.numel(), self.state, float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                 skip if skip is not False else None, skip.numel() // 2 if skip is not False else 0, self._ticket)
        # the update went through raw pointers (no tensor._version bump): tell the net to drop derived copies of its weights
        changed = getattr(f.module, "weights_changed", None)
        if changed is not None:
            changed()

    def zero_grad(self):
        pass  # every backward overwrites the flat gradient buffer

    def layout(self):
        """The fingerprint of the flat layout: (parameter name, offset, numel) in buffer order.  The order inside the flat buffers
        is the net's own business (`flat_param_order`) and has changed between rounds, so a saved optimizer state carries it."""
        f = self._ensure()
        names = {id(p): n for n, p in f.module.named_parameters()}
        return [(names[id(p)], int(off), int(p.numel())) for p, off in zip(f.params, f.offsets)]

    def state_dict(self):
        """Moments and step counters (host copies) + hyper-parameters + the layout the moments are stored in: everything a
        bit-exact resume needs, also after the flat layout of the net has changed."""
        self._ensure()
        return {"m": self.m.cpu(), "v": self.v.cpu(), "state": self.state.cpu(), "lr": self.lr, "betas": tuple(self.betas),
                "eps": self.eps, "weight_decay": self.weight_decay, "layout": self.layout()}

    def load_state_dict(self, sd):
        """Moments are matched to parameters BY NAME through the saved layout; a state saved under another flat order lands on
        the right parameters.  A state without a layout (written before r03) is only accepted where no ambiguity exists: the net
        has no layout of its own (registration order then and now).  Anything else is refused -- equal element counts say
        nothing about where a parameter's moments are."""
        f = self._ensure()
        cur = self.layout()
        saved = sd.get("layout")
        if saved is None:
            reg, off = [], 0
            names = {id(p): n for n, p in f.module.named_parameters()}
            for p in f.module.parameters():
                reg.append((names[id(p)], off, p.numel()))
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if reg != cur or sd["m"].numel() != f.flat_p.numel():
                raise ValueError("optimizer state carries no layout fingerprint and this net orders its flat buffer itself "
                                 "(flat_param_order): the saved Adam moments cannot be matched to parameters; re-save the "
                                 "training state with this build or resume from the model weights alone")
            saved = cur
        saved = [(str(n), int(o), int(k)) for n, o, k in saved]
        if sorted((n, k) for n, _, k in saved) != sorted((n, k) for n, _, k in cur):
            missing = sorted({n for n, _, _ in cur} ^ {n for n, _, _ in saved})
            raise ValueError("optimizer state belongs to another net: parameters differ (%s%s)" %
                             (", ".join(missing[:4]) or "shapes", " ..." if len(missing) > 4 else ""))
        if saved == cur and sd["m"].numel() == f.flat_p.numel():
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        else:
            where = {n: o for n, o, _ in saved}
            m_new, v_new = torch.zeros(f.flat_p.numel()), torch.zeros(f.flat_p.numel())
            for n, o, k in cur:
                so = where[n]
                m_new[o:o + k] = sd["m"][so:so + k]
                v_new[o:o + k] = sd["v"][so:so + k]
            self.m.copy_(m_new); self.v.copy_(v_new)
        self.state.copy_(sd["state"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
