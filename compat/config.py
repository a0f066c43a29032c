"""reference config.py (path constants) -> jlm_amd.config; root from $JLM_ROOT."""
from jlm_amd.config import ExperimentConfig, get_configs  # noqa: F401
from jlm_amd import config as _c

root_path = _c.root_path
train_path = _c.train_path
data_path = _c.data_path
experiment_path = _c.experiment_path
print("root path of project: {}".format(root_path))
