"""reference decoder/model.py -> jlm_amd.model"""
from jlm_amd.model import LSTM_Model  # noqa: F401
