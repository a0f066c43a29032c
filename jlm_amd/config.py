"""Path constants for experiment artefacts (overridable).

Counterpart of the reference's ``config.py`` (reference config.py:14-30): it
exposes the same four names -- ``root_path``, ``train_path``, ``data_path``,
``experiment_path`` -- plus ``ExperimentConfig`` / ``get_configs``.  Unlike the
reference (whose constants are frozen from ``__file__`` at import time) the
root can be moved with :func:`set_root` or the ``JLM_ROOT`` environment
variable, because synthetic fixtures live in scratch directories.

Directory layout under the root (same as the reference):

    data/lexicon.pkl, data/reading_dict.pkl, data/test.txt
    train/experiments/<id>/config.json
    train/experiments/<id>/weights/lstm_weights.pkl
"""
import json
import os

root_path = os.environ.get("JLM_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_artifacts"))
train_path = os.path.join(root_path, "train")
data_path = os.path.join(root_path, "data")
experiment_path = os.path.join(train_path, "experiments")


def set_root(path):
    """Re-point every path constant at ``path`` (reference config.py:14-19)."""
    global root_path, train_path, data_path, experiment_path
    root_path = os.path.abspath(path)
    train_path = os.path.join(root_path, "train")
    data_path = os.path.join(root_path, "data")
    experiment_path = os.path.join(train_path, "experiments")
    return root_path


class ExperimentConfig:
    """Attribute view over a config.json dict (reference config.py:21-26)."""

    def __init__(self, **entries):
        self.__dict__.update(entries)

    def __repr__(self):
        return str(self.__dict__)


def load_config_dict(experiment):
    with open(os.path.join(experiment_path, str(experiment), "config.json"), "rt") as f:
        return json.loads(f.read())


def get_configs(experiment):
    """reference config.py:28-30."""
    return ExperimentConfig(**load_config_dict(experiment))
