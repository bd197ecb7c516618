"""Independent torch-CPU (autograd) mirror of the CRNN graph used to cross-check the hand-written backward of oracle/:
it lives in oracle/torch_port.py (bench.py's CPU-baseline leg times the same code in float32); re-exported here for the tests."""
from oracle.torch_port import *  # noqa: F401,F403
from oracle.torch_port import t, forward, ctc_cost, sampler  # noqa: F401
