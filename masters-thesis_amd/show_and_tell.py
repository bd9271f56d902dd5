"""ShowAndTell (BASELINE config 1) descriptors: ``ShowAndTell/model.py`` Encoder (10-20:
Dense + relu), Decoder (23-65: masked Embedding, LSTM, linear fc1 -> fc2) and the
``CaptionGenerator`` train step (125-164: loss over i = 1..T-1, gradient of the un-normalised
sum).  The arithmetic is ``think_and_tell.CaptionGenerator`` with ``show_and_tell`` set."""
from . import think_and_tell as _tt


class Encoder(_tt.Encoder):
    def __init__(self, embedding_dim):
        super().__init__(embedding_dim)
        self.show_and_tell = True


class Decoder(_tt.Decoder):
    def __init__(self, embedding_dim, units, vocab_size, use_stateful=False):
        super().__init__(embedding_dim, units, vocab_size)
        self.show_and_tell = True


CaptionGenerator = _tt.CaptionGenerator
