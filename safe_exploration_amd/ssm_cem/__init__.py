from .ssm_cem import CemSSM
from .gp_ssm_cem import GpCemSSM

__all__ = ['CemSSM', 'GpCemSSM']
