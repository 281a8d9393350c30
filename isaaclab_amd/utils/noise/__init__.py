from .noise_cfg import (ConstantNoiseCfg, GaussianNoiseCfg, NoiseCfg, NoiseModelCfg, NoiseModelWithAdditiveBiasCfg,
                        UniformNoiseCfg)
from .noise_model import NoiseModel, NoiseModelWithAdditiveBias, constant_noise, gaussian_noise, uniform_noise

# aliases used by task configs (LAB/utils/noise/__init__.py:34-35)
ConstantBiasNoiseCfg = ConstantNoiseCfg
AdditiveUniformNoiseCfg = UniformNoiseCfg
AdditiveGaussianNoiseCfg = GaussianNoiseCfg
