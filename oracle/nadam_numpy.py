"""fp64 numpy restatement of Keras `Nadam` (optimizer_v2, TF 2.1 era) -- TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.

The reference instantiates it at train.py:79-81 (`Nadam(learning_rate=config['learning_rate'])`) and applies it at
models/trainClass.py:132.  TensorFlow is not installable here; the update rule is restated from the Keras documentation /
source semantics summarised in SURVEY.md A.5 (defaults beta_1 0.9, beta_2 0.999, epsilon 1e-7, schedule_decay 0.004,
momentum cache = running product of mu_t) and cross-checked against torch.optim.NAdam(momentum_decay=0.004, eps=1e-7),
which is algebraically the same rule.
"""
import numpy as np


class Nadam:
    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, schedule_decay=0.004):
        self.lr, self.b1, self.b2, self.eps, self.decay = lr, beta_1, beta_2, epsilon, schedule_decay
        self.t, self.m_schedule, self.m, self.v = 0, 1.0, None, None

    def step(self, theta, g):
        theta, g = np.asarray(theta, np.float64), np.asarray(g, np.float64)
        if self.m is None:
            self.m, self.v = np.zeros_like(theta), np.zeros_like(theta)
        self.t += 1
        t = self.t
        mu_t = self.b1 * (1.0 - 0.5 * 0.96 ** (t * self.decay))
        mu_t1 = self.b1 * (1.0 - 0.5 * 0.96 ** ((t + 1) * self.decay))
        m_schedule_new = self.m_schedule * mu_t
        m_schedule_next = m_schedule_new * mu_t1
        self.m_schedule = m_schedule_new
        g_prime = g / (1.0 - m_schedule_new)
        self.m = self.b1 * self.m + (1.0 - self.b1) * g
        m_prime = self.m / (1.0 - m_schedule_next)
        self.v = self.b2 * self.v + (1.0 - self.b2) * g * g
        v_prime = self.v / (1.0 - self.b2 ** t)
        m_bar = (1.0 - mu_t) * g_prime + mu_t1 * m_prime
        return theta - self.lr * m_bar / (np.sqrt(v_prime) + self.eps)


class Adam:
    """Keras `Adam` (optimizer_v2, amsgrad=False; train.py:77-78): lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t),
    theta -= lr_t * m / (sqrt(v) + epsilon)  -- epsilon is NOT bias-corrected (unlike torch.optim.Adam)."""

    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.t, self.m, self.v = 0, None, None

    def step(self, theta, g):
        theta, g = np.asarray(theta, np.float64), np.asarray(g, np.float64)
        if self.m is None:
            self.m, self.v = np.zeros_like(theta), np.zeros_like(theta)
        self.t += 1
        self.m = self.b1 * self.m + (1.0 - self.b1) * g
        self.v = self.b2 * self.v + (1.0 - self.b2) * g * g
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        return theta - lr_t * self.m / (np.sqrt(self.v) + self.eps)


class SGD:
    """Keras `SGD(learning_rate)` without momentum (train.py:82-83)."""

    def __init__(self, lr=1e-2):
        self.lr = lr

    def step(self, theta, g):
        return np.asarray(theta, np.float64) - self.lr * np.asarray(g, np.float64)
