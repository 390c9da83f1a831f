"""ribodetector_amd: MI355X-native implementation of RiboDetector's batched BiLSTM
inference path (reference: ribodetector/detect.py -> model/model.py -> data_loader/seq_encoder.py)."""
__version__ = "0.1.0"
