"""Root directories and the PATCH_CONFIG registry, same names and values as the
reference's src/configs.py:1-44 (tile size / overlap per model family; list
entries are selected by utils.get_patch_config)."""
ROOT_DATASET_DIR = 'datasets'
ROOT_WEIGHTS_DIR = 'weights'
ROOT_RESULTS_DIR = 'results'


def _cfg(size, overlap):
    return {'patch_size': size, 'patch_overlap': overlap}


PATCH_CONFIG = {
    'REDNet': _cfg(128, 32),
    'DnCNN': _cfg(256, 48),
    'DeblurGANv2': [_cfg(768, 128), _cfg(2048, 384)],     # [Inception, MobileNet]
    'Restormer': [_cfg(256, 48), _cfg(512, 96)],          # [denoising, deblurring]
    'MaIR': [_cfg(128, 32), _cfg(384, 128)],              # [gaussian, other]
}
