"""keras.optimizers.Adam / SGD look-alikes (train.py:187-190) whose update runs as one fused HIP kernel
over the flat parameter buffer (global-norm clip + update), Keras 2.2.2 formulas."""


class Optimizer:
    def apply(self, engine, iteration):
        raise NotImplementedError


class Adam(Optimizer):
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, clipnorm=None, **kwargs):
        self.lr, self.beta_1, self.beta_2 = lr, beta_1, beta_2
        self.epsilon = 1e-7 if epsilon is None else epsilon   # K.epsilon()
        self.decay, self.clipnorm = decay, clipnorm

    def apply(self, engine, iteration):
        lr = self.lr / (1.0 + self.decay * iteration) if self.decay > 0 else self.lr
        engine.adam_step(lr, self.beta_1, self.beta_2, self.epsilon, self.clipnorm, iteration)


class SGD(Optimizer):
    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False, clipnorm=None, **kwargs):
        self.lr, self.momentum, self.decay, self.nesterov, self.clipnorm = lr, momentum, decay, nesterov, clipnorm

    def apply(self, engine, iteration):
        engine.sgd_step(self.lr, self.decay, self.momentum, self.nesterov, self.clipnorm, iteration)
