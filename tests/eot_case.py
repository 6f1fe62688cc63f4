"""Test models whose greedy loop ENDS ON eot (Whisper.cpp:219 `while (idx != WHISPER_EOT && ...)`), at a different step
for every clip of a batch.

Seeded synthetic weights never emit eot on their own, so the model is shaped until they do — both sides (oracle and
engine) read the same weights file, so this is still a parity test of the loop, not of the weights:

  * coordinate 0 of the decoder's residual stream is set aside for the stop signal: column 0 of the token embedding is
    zero for every id except eot, so that coordinate reaches no logit but eot's (the output projection is the tied
    embedding, export_onnx.py:378-385);
  * the positional embedding adds a ramp along that coordinate (the signal rises with every decoded id) and row 0 of
    every cross-attention output projection is scaled up (the signal depends on the audio, so clips of one batch cross
    at different steps); all cross-attention output projections are scaled a little so that trajectories differ by clip;
  * logit[eot] = g * (ln(x)[0] + beta0): g and the final LayerNorm's bias beta0 are chosen by `calibrate` from one oracle
    pass with eot disabled (ids before the stop do not depend on them) so that the stops spread over the budget and
    every step up to and including the stop keeps a margin between logit[eot] and the best other logit.

The expected ids/stops are ALWAYS those of an oracle run on the final weights (`EotCase.expect`), never the
calibration's prediction."""
import numpy as np

import modelgen
import oracle


def _round16(x, dtype):
    import torch

    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(torch.float16 if dtype == "F16" else torch.bfloat16).to(torch.float32).numpy()


def eot_clips(n):
    """n seeded clips that differ in everything the log-mel keeps: length (0.5-30 s: the zero-filled tail of the
    features, Whisper.cpp:169-174), noise floor, a tone or a chirp anywhere between 80 Hz and 7 kHz, a silent stretch.
    The cross-attention averages, and with them the stop signal, move from clip to clip."""
    out = []
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64(7000 + i))
        ns = int(rng.integers(8000, 480001))
        t = np.arange(ns, dtype=np.float64) / 16000.0
        x = rng.standard_normal(ns) * 10.0 ** rng.uniform(-4.0, -1.0)
        f0, f1 = rng.uniform(80.0, 7000.0, size=2)
        if i % 2:
            f1 = f0
        x += rng.uniform(0.05, 0.5) * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / max(t[-1], 1e-3)))
        if i % 3 == 0:
            a, b = sorted(rng.integers(0, ns, size=2))
            x[a:b] *= 1e-3
        out.append(np.clip(x, -1.0, 1.0).astype(np.float32))
    return out


class EotCase:
    def __init__(self, model_type, seed, dtype="BF16", cross_scale=8.0, row0_scale=8.0, ramp=4.0, budget=56, n_cal=16,
                 min_margin=0.02, base="benign", gains=(0.5, 1.0, 2.0, 4.0, 8.0)):
        """base "realistic": the stop signal rides on modelgen.realistic_weights (outlier channels, saturated attention,
        logit std ~10: the other logits' maximum is ~50-100 and moves from step to step) instead of N(0, 0.02) weights."""
        self.model_type, self.dtype, self.budget, self.gains, self.base = model_type, dtype, budget, gains, base
        self.dims = modelgen.DIMS[model_type]
        self.cfg = modelgen.make_config(model_type, self.dims)
        self.eot = int(self.cfg["eot"])
        self.policy = 2 if dtype == "F16" else True
        if base == "realistic":
            w = dict(modelgen.realistic_weights(self.dims, seed, dtype=dtype))
        else:
            w = dict(modelgen.synth_weights(self.dims, seed, bf16=(dtype != "F16")))
        if dtype == "F16":
            w = {k: v.astype(np.float16).astype(np.float32) for k, v in w.items()}
        for l in range(self.dims["dec_layers"]):
            o = w[f"decoder.blocks.{l}.cross_attn.out.weight"] * np.float32(cross_scale)   # powers of two: still 16-bit values
            o[0] *= np.float32(row0_scale)
            w[f"decoder.blocks.{l}.cross_attn.out.weight"] = o
        emb = w["decoder.token_embedding.weight"].copy()
        emb[:, 0] = 0.0
        w["decoder.token_embedding.weight"] = emb
        pe = w["decoder.positional_embedding"].copy()
        pe[:, 0] = _round16(pe[:, 0] + ramp * np.arange(pe.shape[0], dtype=np.float32) / 64.0, dtype)
        w["decoder.positional_embedding"] = pe
        self.weights = w
        self.g, self.beta0, self.predicted = self._calibrate(n_cal, min_margin)
        emb[self.eot] = 0.0   # logit[eot] = g * ln(x)[0] and nothing else
        emb[self.eot, 0] = _round16(np.float32([self.g]), dtype)[0]
        lb = w["decoder.ln.bias"].copy()
        lb[0] = _round16(np.float32([self.beta0]), dtype)[0]
        w["decoder.ln.bias"] = lb
        self.oracle = oracle.Oracle(self.cfg, w, bf16_policy=self.policy)
        self._expect = {}

    def _calibrate(self, n_cal, min_margin):
        """One oracle pass over n_cal clips with eot disabled -> h0[b][j] (the stop signal before gain and bias) and the
        best other logit; then a small grid over (g, beta0): most distinct stops, all inside the budget, margin kept."""
        # the signal is read through eot's own logit: a 2^-10 entry in its row makes logit[eot] = h0 / 1024, which never
        # wins (|h0| <= sqrt(d)), and eot = -1 keeps the loop running over the whole budget
        probe = dict(self.weights)
        pemb = probe["decoder.token_embedding.weight"].copy()
        pemb[self.eot] = 0.0
        pemb[self.eot, 0] = 2.0 ** -10
        probe["decoder.token_embedding.weight"] = pemb
        orc = oracle.Oracle(self.cfg, probe, bf16_policy=self.policy)
        h0, mx = [], []
        for c in eot_clips(n_cal):
            mel, _, _ = oracle.log_mel(c, self.dims["n_mels"])
            ck, cv = orc.encoder(mel)
            ids, lg = orc.greedy(ck, cv, "zh", max_new=self.budget, want_logits=True, eot=-1)
            assert self.eot not in ids
            h0.append(lg[:, self.eot].astype(np.float64) * 1024.0)
            mx.append(np.delete(lg, self.eot, axis=1).max(axis=1).astype(np.float64))
        h0, mx = np.array(h0), np.array(mx)       # [n_cal, budget + 1]
        best = None
        lo, hi = float(h0.min()), float(h0.max())
        for g in self.gains:
            for beta0 in np.linspace(-hi, 2.0 / g - lo, 400):
                le = float(_round16(np.float32([g]), self.dtype)[0]) * (h0 + float(_round16(np.float32([beta0]), self.dtype)[0]))
                stops, marg = [], 1e9
                for b in range(len(h0)):
                    hit = np.nonzero(le[b] > mx[b])[0]
                    k = int(hit[0]) if len(hit) else self.budget
                    stops.append(k)
                    marg = min(marg, float(np.abs(le[b, : min(k, self.budget - 1) + 1] - mx[b, : min(k, self.budget - 1) + 1]).min()))
                if marg < min_margin or min(stops) < 1:
                    continue
                n_eot = sum(1 for k in stops if k < self.budget)
                score = (n_eot, len(set(stops)), marg)
                if best is None or score > best[0]:
                    best = (score, g, float(beta0), stops)
        assert best is not None, "no (gain, bias) keeps the margin: change ramp / row0_scale"
        return best[1], best[2], best[3]

    def expect(self, i):
        """(ids, margin) of the oracle on clip i of eot_clips with the final weights; the run ends on eot or on the
        budget. margin: over every step of the run (the one that emits eot included), the smaller of |logit[eot] - best
        other logit| and 5x the gap between the two best other logits — logit[eot] carries the gain g, so a numerical
        difference on the GPU side shows up g times larger in it than in an ordinary logit."""
        if i not in self._expect:
            c = eot_clips(i + 1)[i]
            mel, _, _ = oracle.log_mel(c, self.dims["n_mels"])
            ck, cv = self.oracle.encoder(mel)
            ids, lg = self.oracle.greedy(ck, cv, "zh", max_new=self.budget, want_logits=True)
            top2 = np.partition(np.delete(lg, self.eot, axis=1), -2, axis=1)[:, -2:]
            m_eot = float(np.abs(lg[:, self.eot] - top2[:, 1]).min())
            # realistic base: the vocabulary holds near-duplicate rows on purpose (gaps of 0 .. 1e-2 between the two best
            # text ids: the tie rule's business), so only the stop decision's own margin is required there
            self._expect[i] = (ids, m_eot if self.base == "realistic" else min(m_eot, 5.0 * float((top2[:, 1] - top2[:, 0]).min())))
        return self._expect[i]

    def select(self, n, min_margin=0.02):
        """The first n clips of eot_clips whose oracle run keeps `min_margin` at every step (the calibration predicts the
        margins before the final LayerNorm's 16-bit rounding; this is the run on the final weights) -> (clip index list,
        clips, expected ids). At least one clip must stop on eot and the stops must not all be equal."""
        idx, i = [], 0
        while len(idx) < n:
            assert i < 4 * n + 16, "too few clips keep the margin"
            if self.expect(i)[1] >= min_margin:
                idx.append(i)
            i += 1
        clips = eot_clips(max(idx) + 1)
        ids = [self.expect(i)[0] for i in idx]
        assert any(len(x) < self.budget for x in ids)
        return idx, [clips[i] for i in idx], ids

    def write(self, root):
        modelgen.write_model_dir(str(root), self.model_type, self.dims, weights=self.weights, dtype=self.dtype)
        return str(root)
