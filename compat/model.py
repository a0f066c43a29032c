"""reference decoder/model.py -> jlm_amd.model"""
from jlm_amd.model import LSTM_Model, sigmoid, softmax, tanh, find_top_N, sample  # noqa: F401
