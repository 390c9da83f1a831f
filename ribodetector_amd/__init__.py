"""ribodetector_amd: MI355X-native implementation of RiboDetector's batched BiLSTM
inference path (reference: ribodetector/detect.py -> model/model.py -> data_loader/seq_encoder.py)."""
__version__ = "0.1.0"

import os as _os

# ROCclr keeps a pool of completion signals per queue (ROC_SIGNAL_POOL_SIZE, default 64). The CLI's pipeline has hundreds of small
# commands in flight (index / gather / pack / copy kernels of several streams beside the recurrence launches); with the default pool the
# runtime's helper thread recycles signals in a busy loop - ~0.6 s of CPU per second of GPU work, measured on MI355X / ROCm 7.2
# (tools/scratch/env_ab.sh: BGZF -> gz at 1.30 host cores with 64, 0.75 with 256 ... 65536). It is read when the HIP runtime
# initialises - the first GPU call of the process - so importing this package before that is early enough; an explicit setting wins.
_os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "1024")
