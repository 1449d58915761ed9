"""train.py imports Plotter; plotting is out of scope for the hot path (flair/visual OUT OF SCOPE)."""


class Plotter(object):
    def plot_training_curves(self, *a, **k):
        pass

    def plot_weights(self, *a, **k):
        pass

    def plot_learning_rate(self, *a, **k):
        pass
