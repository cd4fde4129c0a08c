"""CPU restatement (numpy) of the reference's CTC head and greedy decode.  TEST INFRASTRUCTURE ONLY.

Reference: src/models/asr_model.py:34-50 (CTCDecoder.forward: linear projection to the
vocabulary, log-softmax over it), :158-187 (decode: per utterance, the first input_lengths[i]
frames) and :189-213 (_greedy_decode: argmax per frame, drop blanks and consecutive
repeats; blank = vocab_size - 1, :204).  Pinned by tests/golden/ctc_greedy.npz, generated from the
reference's own classes by tests/golden/make_golden_ctc.py.
"""
import numpy as np


def log_probs(enc, W, b):
    """[B, T, Hd] x [V, Hd] -> log-softmax logits [B, T, V] (src/models/asr_model.py:45-48)."""
    z = np.asarray(enc, dtype=np.float64) @ np.asarray(W, dtype=np.float64).T + np.asarray(b, dtype=np.float64)
    z = z - z.max(axis=-1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def greedy_decode(lp, lengths, blank=None):
    """List of token lists (src/models/asr_model.py:171-213)."""
    lp = np.asarray(lp)
    blank = lp.shape[-1] - 1 if blank is None else blank
    out = []
    for i in range(lp.shape[0]):
        best = lp[i, : int(lengths[i])].argmax(axis=-1)  # first maximum, as torch.argmax on the CPU
        seq, prev = [], None
        for tok in best.tolist():
            if tok != blank and tok != prev:
                seq.append(tok)
            prev = tok
        out.append(seq)
    return out


def _lse(*xs):
    m = max(xs)
    if m == -np.inf:
        return -np.inf
    return m + np.log(sum(np.exp(x - m) for x in xs))


def ctc_loss_and_grad(lp, targets, input_lengths, target_lengths, blank=None, zero_infinity=True):
    """Mean-reduced CTC loss and the gradient torch returns for log_probs, by the textbook alpha-beta recursion in
    float64.  lp [B, T, V] (batch first), targets [B, Lmax] padded.  What nn.CTCLoss(blank=V-1, reduction='mean',
    zero_infinity=True) computes in the reference (src/models/asr_model.py:105, 144-154); its arithmetic lives in
    torch, to which tests/test_ctc.py pins this restatement (loss and gradient).  The gradient is the one through
    the log-softmax that produced lp: grad[t, c] = w (exp lp[t, c] - occupancy[t, c]), rows summing to zero."""
    lp = np.asarray(lp, dtype=np.float64)
    B, T, V = lp.shape
    blank = V - 1 if blank is None else blank
    grad = np.zeros_like(lp)
    per = np.zeros(B)
    for b in range(B):
        Tb, L = int(input_lengths[b]), int(target_lengths[b])
        lab = [int(x) for x in targets[b][:L]]
        ext = [blank]
        for c in lab:
            ext += [c, blank]
        S = len(ext)
        alpha = np.full((Tb, S), -np.inf)
        beta = np.full((Tb, S), -np.inf)
        if Tb == 0:
            per[b] = 0.0 if zero_infinity else np.inf
            continue
        alpha[0, 0] = lp[b, 0, blank]
        if S > 1:
            alpha[0, 1] = lp[b, 0, ext[1]]
        for t in range(1, Tb):
            for s in range(S):
                terms = [alpha[t - 1, s]]
                if s >= 1:
                    terms.append(alpha[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    terms.append(alpha[t - 1, s - 2])
                alpha[t, s] = _lse(*terms) + lp[b, t, ext[s]]
        ll = _lse(alpha[Tb - 1, S - 1], alpha[Tb - 1, S - 2] if S > 1 else -np.inf)
        nll = -ll
        w = 1.0 / (B * max(L, 1))
        if not np.isfinite(nll):
            per[b] = 0.0 if zero_infinity else nll
            continue
        per[b] = nll / max(L, 1)
        beta[Tb - 1, S - 1] = lp[b, Tb - 1, blank]
        if S > 1:
            beta[Tb - 1, S - 2] = lp[b, Tb - 1, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            for s in range(S):
                terms = [beta[t + 1, s]]
                if s + 1 < S:
                    terms.append(beta[t + 1, s + 1])
                if s + 2 < S and ext[s] != blank and ext[s] != ext[s + 2]:
                    terms.append(beta[t + 1, s + 2])
                beta[t, s] = _lse(*terms) + lp[b, t, ext[s]]
        for t in range(Tb):
            occ = np.full(V, -np.inf)
            for s in range(S):
                occ[ext[s]] = _lse(occ[ext[s]], alpha[t, s] + beta[t, s])
            grad[b, t] = w * (np.exp(lp[b, t]) - np.exp(occ + nll - lp[b, t]))
    return per.mean(), grad
