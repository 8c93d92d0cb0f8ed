from basicsr.utils.registry import ARCH_REGISTRY
