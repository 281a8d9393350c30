from .frame_transformer import FrameTransformer, FrameTransformerCfg, FrameTransformerData

__all__ = ["FrameTransformer", "FrameTransformerCfg", "FrameTransformerData"]
