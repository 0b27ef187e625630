"""anatomask_amd: MI355X-native implementation of AnatoMask's masked-image-modelling pretraining step."""
__version__ = "0.1.0"


def set_f32_products(mode: str) -> None:
    """What fp32-storage models (`compute_dtype=torch.float32`) CONSTRUCTED AFTER this call start with: "exact" = the exact fp32 matrix
    instruction (parity mode, default), "split" = products from bf16 hi / lo splits of both operands with fp32 accumulation (AM_DT_F32S:
    16 significant bits per operand, 4x the matrix rate in the convolutions).  The mode itself is a property of each model
    (`SparK.set_f32_split`, `AnatoMaskTrainer(f32_split=True)`): existing models are not touched."""
    from . import ops
    if mode not in ("exact", "split"):
        raise ValueError(mode)
    ops.DEFAULT_F32_SPLIT = mode == "split"
