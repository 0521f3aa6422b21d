from .test import calculate_uncertainty, single_gpu_test, single_gpu_uncertainty
from .train_Lambda import train_detector_SSL

__all__ = ['calculate_uncertainty', 'single_gpu_test', 'single_gpu_uncertainty', 'train_detector_SSL']
