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
