"""ms2_NIC -- multi-subject model (BASELINE config 5).

Drop-in for ``AttemptFour/Model/ms2_NIC.py`` (class NIC, lines 37-465): same constructor
(ms2_NIC.py:46), one region-wise encoder per subject on equal batch slices, shared
attention / LSTM / head, ``train_step`` returning ``loss`` (mean of the per-subject
cross-entropies, ms2_NIC.py:355) plus ``lossA/B``, ``accuracyA/B``, ``attentionA/B``.
The reference hard-codes 2 subjects (``dense_in_a`` / ``dense_in_b``); ``n_subjects`` generalises
it.  The data generator stacks the subjects' batches (data_generator_multisub.py:77-102), so the
model batch is n_subjects * batch_size, split in equal slices exactly as ms2_NIC.call does
(ms2_NIC.py:181-188).
"""
from . import lc_nic


class NIC(lc_nic.NIC):
    def __init__(self, groups, units, embedding_features, embedding_text, attn_units, vocab_size, max_length,
                 dropout_input, dropout_features, dropout_text, dropout_attn, dropout_lstm, dropout_out, input_reg,
                 attn_reg, lstm_reg, output_reg, n_subjects=2, **kw):
        super().__init__(groups, units, embedding_features, embedding_text, attn_units, vocab_size, max_length,
                         dropout_input, dropout_features, dropout_text, dropout_attn, dropout_lstm, dropout_out,
                         input_reg, attn_reg, lstm_reg, output_reg, n_subjects=n_subjects, **kw)

    def __call__(self, data, training=True):
        """ms2_NIC.call (ms2_NIC.py:177-205) returns (predictionA, attention_scoresA, predictionB,
        attention_scoresB, ...); sub-calls always run with training=True (quirk kept)."""
        probs, alpha = super().__call__(data, training=True)
        Bs = probs.shape[0] // self.S
        out = []
        for q in range(self.S):
            out += [probs[q * Bs:(q + 1) * Bs], alpha[:, q * Bs:(q + 1) * Bs]]
        return tuple(out)

    call = __call__
