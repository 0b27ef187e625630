"""anatomask_amd: MI355X-native implementation of AnatoMask's masked-image-modelling pretraining step."""
__version__ = "0.1.0"
