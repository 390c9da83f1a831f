"""float64 evaluation of the reference's forward1 (model/model.py:32-37: BiLSTM over the packed reads, hidden state at the
last base, Linear) - the yardstick for how far ANY fp32 implementation (the reference's own torch/cuDNN arithmetic, the
oracle, the HIP kernels) is from the exact value of the same recurrence. Test helper, numpy only."""
import numpy as np


def f64_forward(sd, arena, off, lens, max_len):
    """float64 evaluation of forward1 (packed semantics), vectorised over reads"""
    g = lambda k: np.asarray(sd[k], dtype=np.float64) if isinstance(sd[k], np.ndarray) else sd[k].double().numpy()   # noqa: E731
    wih, whh, b = g("rnn.weight_ih_l0"), g("rnn.weight_hh_l0"), g("rnn.bias_ih_l0") + g("rnn.bias_hh_l0")
    wihr, br = g("rnn.weight_ih_l0_reverse"), g("rnn.bias_ih_l0_reverse") + g("rnn.bias_hh_l0_reverse")
    wout, bout = g("out.weight"), g("out.bias")
    n = len(lens)
    T = np.minimum(lens, max_len).astype(np.int64)
    lut = np.full(256, 4, dtype=np.int64)
    for ch, c in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"U", 3)):
        lut[ch[0]] = c
    inl = np.concatenate([wih.T + b, b[None, :]], 0)                # [5, 512]
    inr = np.concatenate([wihr.T + br, br[None, :]], 0)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))                         # noqa: E731
    h = np.zeros((n, 128)); c = np.zeros((n, 128))
    order = np.argsort(-T, kind="stable")
    Ts = T[order]
    for t in range(int(T.max()) if n else 0):
        m = int((Ts > t).sum())
        idx = order[:m]
        code = lut[arena[off[idx] + t]]
        gates = inl[code] + h[idx] @ whh.T
        i, f, gg, o = sig(gates[:, :128]), sig(gates[:, 128:256]), np.tanh(gates[:, 256:384]), sig(gates[:, 384:])
        c[idx] = f * c[idx] + i * gg
        h[idx] = o * np.tanh(c[idx])
    last = np.where(T > 0, lut[arena[np.minimum(off[:-1] + np.maximum(T, 1) - 1, len(arena) - 1)]], 4)
    gr = inr[last]
    cr = sig(gr[:, :128]) * np.tanh(gr[:, 256:384])
    hr = sig(gr[:, 384:]) * np.tanh(cr)
    hr[T == 0] = 0.0
    return np.concatenate([h, hr], 1) @ wout.T + bout




def f64_forward_torch(sd, arena, L, dev, block=1 << 18):
    """The same float64 evaluation for fixed-length reads on a torch device (the GPU's float64 units make 10^6-10^7 reads a matter
    of seconds): arena uint8[n*L] on `dev` -> logits float64 [n, 2]."""
    import torch
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float64, device=dev)   # noqa: E731
    wih, whh, b = g("rnn.weight_ih_l0"), g("rnn.weight_hh_l0"), g("rnn.bias_ih_l0") + g("rnn.bias_hh_l0")
    wihr, br = g("rnn.weight_ih_l0_reverse"), g("rnn.bias_ih_l0_reverse") + g("rnn.bias_hh_l0_reverse")
    wout, bout = g("out.weight"), g("out.bias")
    lut = torch.full((256,), 4, dtype=torch.int64, device=dev)
    for ch, c in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"U", 3)):
        lut[ch[0]] = c
    inl = torch.cat([wih.T + b, b[None, :]], 0)
    inr = torch.cat([wihr.T + br, br[None, :]], 0)
    out = []
    reads = arena.view(-1, L)
    for s in range(0, reads.shape[0], block):
        code = lut[reads[s:s + block].long()]
        n = code.shape[0]
        h = torch.zeros((n, 128), dtype=torch.float64, device=dev)
        c = torch.zeros_like(h)
        for t in range(L):
            gates = inl[code[:, t]] + h @ whh.T
            i, f, gg, o = torch.sigmoid(gates[:, :128]), torch.sigmoid(gates[:, 128:256]), torch.tanh(gates[:, 256:384]), torch.sigmoid(gates[:, 384:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
        gr = inr[code[:, L - 1]]
        cr = torch.sigmoid(gr[:, :128]) * torch.tanh(gr[:, 256:384])
        hr = torch.sigmoid(gr[:, 384:]) * torch.tanh(cr)
        out.append(torch.cat([h, hr], 1) @ wout.T + bout)
    return torch.cat(out, 0)


def f64_forward_torch_varlen(sd, arena, off, lens, max_len, dev, block=1 << 16):
    """float64 forward1 for variable-length reads on a torch device: arena uint8[*], off int64[n+1], lens int32[n] (numpy) ->
    logits float64 [n, 2] (numpy). Reads are gathered into a padded code matrix; a read's state stops changing after its last base."""
    import torch
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float64, device=dev)   # noqa: E731
    wih, whh, b = g("rnn.weight_ih_l0"), g("rnn.weight_hh_l0"), g("rnn.bias_ih_l0") + g("rnn.bias_hh_l0")
    wihr, br = g("rnn.weight_ih_l0_reverse"), g("rnn.bias_ih_l0_reverse") + g("rnn.bias_hh_l0_reverse")
    wout, bout = g("out.weight"), g("out.bias")
    lut = np.full(256, 4, dtype=np.int64)
    for ch, c in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"U", 3)):
        lut[ch[0]] = c
    inl = torch.cat([wih.T + b, b[None, :]], 0)
    inr = torch.cat([wihr.T + br, br[None, :]], 0)
    n = len(lens)
    T = np.minimum(np.asarray(lens, dtype=np.int64), max_len)
    out = np.zeros((n, 2))
    for s in range(0, n, block):
        e = min(n, s + block)
        Tb = T[s:e]
        L = int(Tb.max()) if e > s else 0
        pos = np.arange(L)[None, :]
        idx = np.minimum(off[s:e, None] + pos, len(arena) - 1)
        code = torch.as_tensor(np.where(pos < Tb[:, None], lut[arena[idx]], 4), device=dev)
        Tt = torch.as_tensor(Tb, device=dev)
        h = torch.zeros((e - s, 128), dtype=torch.float64, device=dev)
        c = torch.zeros_like(h)
        for t in range(L):
            gates = inl[code[:, t]] + h @ whh.T
            i, f, gg, o = torch.sigmoid(gates[:, :128]), torch.sigmoid(gates[:, 128:256]), torch.tanh(gates[:, 256:384]), torch.sigmoid(gates[:, 384:])
            c2 = f * c + i * gg
            h2 = o * torch.tanh(c2)
            live = (Tt > t)[:, None]
            c = torch.where(live, c2, c)
            h = torch.where(live, h2, h)
        last = torch.as_tensor(np.where(Tb > 0, lut[arena[np.minimum(off[s:e] + np.maximum(Tb, 1) - 1, len(arena) - 1)]], 4), device=dev)
        gr = inr[last]
        cr = torch.sigmoid(gr[:, :128]) * torch.tanh(gr[:, 256:384])
        hr = torch.sigmoid(gr[:, 384:]) * torch.tanh(cr)
        hr = torch.where((Tt > 0)[:, None], hr, torch.zeros_like(hr))
        out[s:e] = (torch.cat([h, hr], 1) @ wout.T + bout).cpu().numpy()
    return out
