"""Decode / evaluation side of the reference, output-format compatible.

* ``eval_model`` / ``eval_fc_model``  -- AttemptFour/eval.py:148-216: run the generator through
  ``model.greedy_predict`` and write ``output_captions_<epoch>.npy``, ``output_captions_raw_<epoch>.npy``,
  ``attention_scores_<epoch>.npy`` and ``tokenizer.json`` with the reference's array layouts, so the
  thesis' analysis scripts keep working.
* ``ids_to_captions``                 -- tokenizer.sequences_to_texts with the <start>/<end>/<pad> handling of
  ThinkAndTell/evaluate.py:178-201.
* ``sentence_bleu`` / ``bleu_scores`` -- ThinkAndTell/img_evaluate.py:212-250 calls
  nltk.translate.bleu_score.sentence_bleu(references, candidate, weights, SmoothingFunction().method1);
  nltk is not installable here, so the published algorithm (Papineni et al. 2002; Chen & Cherry 2014 method 1:
  a zero n-gram match count becomes epsilon = 0.1) is restated.  PARITY UNPINNED against nltk itself; pinned by
  the worked example of the nltk documentation (tests/test_data_fit.py).
Host-side Python throughout: nothing here is on the per-step hot path.
"""
import math
import os
from collections import Counter

import numpy as np


def eval_model(model, data_generator, tokenizer, config, out_path, epoch, add_name=""):
    """eval.py:148-194.  greedy_predict returns (words (B,T,1), probs (B,T,V), alpha (T,B,R,1), s); the files hold
    outputs (n,T,1), outputs_raw (n,T,V) and attention_scores (n,T,R,1) (eval.py:172-174)."""
    outs, raws, attns = [], [], []
    for i in range(len(data_generator)):
        sample = data_generator[i]
        features, _, a0, c0 = sample[0][:4]
        start_seq = np.repeat([tokenizer.word_index["<start>"]], features.shape[0])
        words, probs, alpha, _ = model.greedy_predict(features, a0, c0, start_seq, config["max_length"], config["units"],
                                                      tokenizer, return_s=False)     # eval.py never reads `s`
        outs.append(words); raws.append(probs); attns.append(alpha)
    outputs = np.concatenate(outs, axis=0)
    outputs_raw = np.concatenate(raws, axis=0)
    attention_scores = np.swapaxes(np.concatenate(attns, axis=1), 0, 1)
    os.makedirs(out_path, exist_ok=True)
    np.save(os.path.join(out_path, f"output_captions_{epoch}{add_name}.npy"), outputs)
    np.save(os.path.join(out_path, f"output_captions_raw_{epoch}{add_name}.npy"), outputs_raw)
    np.save(os.path.join(out_path, f"attention_scores_{epoch}{add_name}.npy"), attention_scores)
    with open(os.path.join(out_path, "tokenizer.json"), "w") as f:
        f.write(tokenizer.to_json())
    return outputs, attention_scores


def eval_fc_model(model, data_generator, tokenizer, config, out_path, epoch, add_name=""):
    """eval.py:196-216: greedy_predict_fc returns ids (T,B,1); the file holds (n,T,1)."""
    outs = []
    for i in range(len(data_generator)):
        sample = data_generator[i]
        features, _, a0, c0 = sample[0][:4]
        start_seq = np.repeat([tokenizer.word_index["<start>"]], features.shape[0])
        outs.append(model.greedy_predict(features, a0, c0, start_seq, config["max_length"], config["units"], tokenizer))
    all_outputs = np.swapaxes(np.concatenate(outs, axis=1), 0, 1)
    os.makedirs(out_path, exist_ok=True)
    np.save(os.path.join(out_path, f"output_captions_{epoch}{add_name}.npy"), all_outputs)
    with open(os.path.join(out_path, "tokenizer.json"), "w") as f:
        f.write(tokenizer.to_json())
    return all_outputs


def ids_to_captions(ids, tokenizer, end_token="<end>", drop=("<start>", "<pad>")):
    """(n,T[,1]) ids -> list of token lists, cut at the first <end> (ThinkAndTell/evaluate.py:178-201); id 0 is
    padding."""
    ids = np.asarray(ids)
    if ids.ndim == 3:
        ids = ids[:, :, 0]
    caps = []
    for row in ids:
        words = []
        for i in row:
            w = tokenizer.index_word.get(int(i)) if int(i) != 0 else None
            if w is None or w in drop:
                continue
            if w == end_token:
                break
            words.append(w)
        caps.append(words)
    return caps


def simple_eval(model, betas, target, tokenizer=None, temperature=1.0, sample_step=0, end_token="<end>"):
    """ThinkAndTell/evaluate.py:261-284 (`simple_eval`): one teacher-forced forward of the caption generator, then one
    categorical draw per position from the logits (tf.random.categorical(logits, 1)); the caption is cut at the first
    <end>.  The draw runs on the device (tnt_sample_rows_f32, Philox stream (seed, S_SAMPLE, sample_step)).
    Returns (ids (B, T+1) int64, captions or None)."""
    import torch
    from . import ops
    from .model_base import S_SAMPLE
    logits = model((betas, None, target), training=False)                 # (B, T+1, V), device tensor
    Bn, Tn, V = logits.shape
    flat = logits.reshape(Bn * Tn, V).contiguous()
    ids = torch.zeros(Bn * Tn, dtype=torch.int32, device=flat.device)
    ops.backend().sample_rows(flat, ids, Bn * Tn, V, V, temperature, True, model.seed, S_SAMPLE, sample_step)
    ids = ids.view(Bn, Tn).cpu().numpy().astype(np.int64)
    caps = None
    if tokenizer is not None:
        caps = []
        for row in ids:
            words = []
            for i in row:
                w = tokenizer.index_word.get(int(i), "<unk>")
                words.append(w)
                if w == end_token:
                    break
            caps.append(words)
    return ids, caps


def _ngrams(seq, n):
    return Counter(tuple(seq[i:i + n]) for i in range(len(seq) - n + 1))


def sentence_bleu(references, hypothesis, weights=(0.25, 0.25, 0.25, 0.25), smoothing="method1", epsilon=0.1):
    """BLEU of one tokenised hypothesis against tokenised references: clipped n-gram precisions, geometric mean
    with ``weights``, brevity penalty against the closest reference length.  smoothing=None: any zero precision
    gives 0; "method1": zero match counts are replaced by ``epsilon`` (Chen & Cherry 2014)."""
    hyp_len = len(hypothesis)
    if hyp_len == 0:
        return 0.0
    p = []
    for n in range(1, len(weights) + 1):
        hyp = _ngrams(hypothesis, n)
        max_ref = Counter()
        for ref in references:
            for g, c in _ngrams(ref, n).items():
                max_ref[g] = max(max_ref[g], c)
        num = sum(min(c, max_ref[g]) for g, c in hyp.items())
        den = max(1, sum(hyp.values()))
        p.append((num, den))
    if p[0][0] == 0:                       # no unigram overlap at all
        return 0.0
    ref_len = min((len(r) for r in references), key=lambda rl: (abs(rl - hyp_len), rl))
    bp = 1.0 if hyp_len > ref_len else math.exp(1.0 - ref_len / hyp_len)
    s = 0.0
    for w, (num, den) in zip(weights, p):
        if num == 0:
            if smoothing is None:
                return 0.0
            num = epsilon
        s += w * math.log(num / den)
    return bp * math.exp(s)


def bleu_scores(references, candidate):
    """The four scores of ThinkAndTell/img_evaluate.py:245-248 (BLEU-1..4, cumulative weights, method 1)."""
    ws = [(1, 0, 0, 0), (0.5, 0.5, 0, 0), (0.33, 0.33, 0.33, 0), (0.25, 0.25, 0.25, 0.25)]
    return tuple(sentence_bleu(references, candidate, weights=w) for w in ws)
