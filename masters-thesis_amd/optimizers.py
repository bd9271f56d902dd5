"""Keras-style optimizer / loss descriptor objects.

They carry hyper-parameters only; the arithmetic is the multi-tensor HIP kernels in
csrc/optim.hip driven by the model (reference construction: AttemptFour/main.py:97-110).
"""


class Adam:
    """tf.keras.optimizers.Adam(learning_rate, beta_1, beta_2, epsilon, clipnorm) -- main.py:97.
    ``clipnorm`` is per variable; None disables it (the TF<=2.3 behaviour of a
    custom tape.gradient -> apply_gradients step)."""

    kind = "adam"

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None, **kw):
        self.lr = float(kw.pop("lr", learning_rate))
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.clipnorm = None if clipnorm is None else float(clipnorm)
        self.iterations = 0

    @property
    def learning_rate(self):
        return self.lr

    @learning_rate.setter
    def learning_rate(self, v):
        self.lr = float(v)


class SGD:
    """tf.keras.optimizers.SGD(learning_rate, momentum, nesterov=False) -- main.py:100-102."""

    kind = "sgd"

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, clipnorm=None, **kw):
        if nesterov:
            raise NotImplementedError("nesterov momentum is not used by the reference path")
        self.lr = float(kw.pop("lr", learning_rate))
        self.momentum = float(momentum)
        self.clipnorm = None if clipnorm is None else float(clipnorm)
        self.iterations = 0

    @property
    def learning_rate(self):
        return self.lr

    @learning_rate.setter
    def learning_rate(self, v):
        self.lr = float(v)


class CategoricalCrossentropy:
    """tf.keras.losses.CategoricalCrossentropy(from_logits=False, reduction='none') -- main.py:107-110.
    Only this configuration is implemented by the fused softmax+CE kernel."""

    def __init__(self, from_logits=False, reduction="none"):
        if from_logits:
            raise NotImplementedError("the reference path uses from_logits=False")
        self.from_logits, self.reduction = from_logits, reduction
