"""Closed-form ordinary least squares without intercept (oracle/shims/README.md)."""
import numpy as np


class _Fit:
    def __init__(self, y, x):
        y = np.asarray(y, dtype=np.float64)
        x = np.asarray(x, dtype=np.float64)
        sxx = float(np.dot(x, x))
        beta = float(np.dot(x, y)) / sxx
        resid = y - beta * x
        rss = float(np.dot(resid, resid))
        dof = len(y) - 1
        self.params = np.array([beta])
        self.bse = np.array([np.sqrt(rss / dof / sxx)])
        self.rsquared = 1.0 - rss / float(np.dot(y, y))
        self._x = x

    def predict(self):
        return self.params[0] * self._x

    def summary(self):
        return "OLS (shim): slope=%r bse=%r R2=%r" % (
            self.params[0],
            self.bse[0],
            self.rsquared,
        )


class OLS:
    def __init__(self, endog, exog):
        self._y, self._x = endog, exog

    def fit(self):
        return _Fit(self._y, self._x)
