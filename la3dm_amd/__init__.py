"""la3dm_amd — MI355X-native (gfx950) implementation of la3dm's per-scan occupancy-inference
hot path behind the reference's BGKOctoMap interface.  See DESIGN.md / INTEGRATION.md."""
from .bgkoctomap import BGKOctoMap, GPOctoMap, BGKLVOctoMap, BGKLOctoMap, PackedScan, FREE, OCCUPIED, UNKNOWN, PRUNED  # noqa: F401
from .pcd import load_pcd  # noqa: F401
from .synth import synthetic_scan  # noqa: F401

BGK_YAML = dict(resolution=0.1, block_depth=3, sf2=1.0, ell=0.2, free_thresh=0.3, occupied_thresh=0.7,
                var_thresh=100.0, prior_A=0.001, prior_B=0.001)  # config/methods/bgkoctomap.yaml
GP_YAML = dict(resolution=0.1, block_depth=3, sf2=1.0, ell=1.0, noise=0.01, l=100.0, min_var=0.001, max_var=1000.0,
               max_known_var=0.02, free_thresh=0.3, occupied_thresh=0.7)  # config/methods/gpoctomap.yaml
LV_YAML = dict(resolution=0.1, block_depth=5, sf2=0.1, ell=0.2, free_thresh=0.3, occupied_thresh=0.7, var_thresh=0.2,
               prior_A=0.001, prior_B=0.001, original_size=True, min_W=0.001)  # config/methods/bgklvoctomap.yaml
L_YAML = dict(resolution=0.1, block_depth=3, sf2=0.1, ell=0.2, free_thresh=0.3, occupied_thresh=0.7, var_thresh=0.15,
              prior_A=0.001, prior_B=0.001)  # config/methods/bgkloctomap.yaml (free_resolution 0.3, ds_resolution 0.1)
