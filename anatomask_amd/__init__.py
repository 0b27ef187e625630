"""anatomask_amd: MI355X-native implementation of AnatoMask's masked-image-modelling pretraining step."""
__version__ = "0.1.0"


def set_f32_products(mode: str) -> None:
    """fp32-storage models (`compute_dtype=torch.float32`): "exact" = the exact fp32 matrix instruction (parity mode, default),
    "split" = products from bf16 hi / lo splits of both operands with fp32 accumulation (AM_DT_F32S: 16 significant bits per operand, 4x the
    matrix rate in the convolutions; `AnatoMaskTrainer(f32_split=True)` sets the same switch).  Call before the first forward of a model (or
    call `model.weights_changed()` afterwards: the packed weight copies are made for one of the two modes)."""
    from . import ops
    if mode not in ("exact", "split"):
        raise ValueError(mode)
    ops.F32_SPLIT = mode == "split"
