"""Parameter groups with the reference's defaults (reference arguments/__init__.py:19-174) and the default
config overlay arguments/dnerf/hellwarrior.py -> dnerf_default.py (applied by `default_config()` because mmcv,
which the reference uses to read those files, is not a dependency here)."""
import os
import sys
from argparse import ArgumentParser, Namespace


class GroupParams:
    pass


class ParamGroup:
    def __init__(self, parser: ArgumentParser, name: str, fill_none=False):
        group = parser.add_argument_group(name)
        for key, value in vars(self).items():
            shorthand = key.startswith("_")
            key = key[1:] if shorthand else key
            t = type(value)
            value = value if not fill_none else None
            flags = ["--" + key] + (["-" + key[0:1]] if shorthand else [])
            if t == bool:
                group.add_argument(*flags, default=value, action="store_true")
            else:
                group.add_argument(*flags, default=value, type=t)

    def extract(self, args):
        g = GroupParams()
        for k, v in vars(args).items():
            if k in vars(self) or ("_" + k) in vars(self):
                setattr(g, k, v)
        return g


class ModelParams(ParamGroup):
    def __init__(self, parser, sentinel=False):
        self.sh_degree = 3
        self._source_path = ""
        self._model_path = ""
        self._images = "images"
        self._resolution = -1
        self._white_background = False
        self.data_device = "cuda"
        self.eval = True
        self.render_process = False
        self.add_points = False
        self.extension = ".png"
        self.llffhold = 8
        super().__init__(parser, "Loading Parameters", sentinel)

    def extract(self, args):
        g = super().extract(args)
        g.source_path = os.path.abspath(g.source_path)
        return g


class PipelineParams(ParamGroup):
    def __init__(self, parser):
        self.convert_SHs_python = False
        self.compute_cov3D_python = False
        self.debug = False
        super().__init__(parser, "Pipeline Parameters")


class ModelHiddenParams(ParamGroup):
    def __init__(self, parser):
        self.net_width = 64
        self.timebase_pe = 4
        self.defor_depth = 1
        self.posebase_pe = 10
        self.scale_rotation_pe = 2
        self.opacity_pe = 2
        self.timenet_width = 64
        self.timenet_output = 32
        self.bounds = 1.6
        self.plane_tv_weight = 0.0001
        self.time_smoothness_weight = 0.01
        self.l1_time_planes = 0.0001
        self.kplanes_config = {'grid_dimensions': 2, 'input_coordinate_dim': 4, 'output_coordinate_dim': 32,
                               'resolution': [64, 64, 64, 25]}
        self.multires = [1, 2, 4, 8]
        self.no_dx = False
        self.no_grid = False
        self.no_ds = False
        self.no_dr = False
        self.no_do = True
        self.no_dshs = True
        self.empty_voxel = False
        self.grid_pe = 0
        self.static_mlp = False
        self.apply_rotation = False
        super().__init__(parser, "ModelHiddenParams")


class OptimizationParams(ParamGroup):
    def __init__(self, parser):
        self.dataloader = False
        self.zerostamp_init = False
        self.custom_sampler = None
        self.iterations = 30_000
        self.coarse_iterations = 3000
        self.position_lr_init = 0.00016
        self.position_lr_final = 0.0000016
        self.position_lr_delay_mult = 0.01
        self.position_lr_max_steps = 20_000
        self.deformation_lr_init = 0.00016
        self.deformation_lr_final = 0.000016
        self.deformation_lr_delay_mult = 0.01
        self.grid_lr_init = 0.0016
        self.grid_lr_final = 0.00016
        self.feature_lr = 0.0025
        self.opacity_lr = 0.05
        self.scaling_lr = 0.005
        self.rotation_lr = 0.001
        self.percent_dense = 0.01
        self.lambda_dssim = 0
        self.lambda_lpips = 0
        self.weight_constraint_init = 1
        self.weight_constraint_after = 0.2
        self.weight_decay_iteration = 5000
        self.opacity_reset_interval = 3000
        self.densification_interval = 100
        self.densify_from_iter = 500
        self.densify_until_iter = 15_000
        self.densify_grad_threshold_coarse = 0.0002
        self.densify_grad_threshold_fine_init = 0.0002
        self.densify_grad_threshold_after = 0.0002
        self.pruning_from_iter = 500
        self.pruning_interval = 100
        self.opacity_threshold_coarse = 0.005
        self.opacity_threshold_fine_init = 0.005
        self.opacity_threshold_fine_after = 0.005
        self.batch_size = 1
        self.add_point = False
        super().__init__(parser, "Optimization Parameters")


# arguments/dnerf/dnerf_default.py:3-33 overlaid by arguments/dnerf/hellwarrior.py:3-10 (train_4DGS.py:432 default)
DEFAULT_CONFIG = dict(
    OptimizationParams=dict(coarse_iterations=3000, deformation_lr_init=0.00016, deformation_lr_final=0.0000016,
                            deformation_lr_delay_mult=0.01, grid_lr_init=0.0016, grid_lr_final=0.000016, iterations=20000,
                            pruning_interval=8000, percent_dense=0.01, render_process=False),
    ModelHiddenParams=dict(multires=[1, 2], defor_depth=0, net_width=64, plane_tv_weight=0.0001, time_smoothness_weight=0.01,
                           l1_time_planes=0.0001, weight_decay_iteration=0, bounds=1.6,
                           kplanes_config={'grid_dimensions': 2, 'input_coordinate_dim': 4, 'output_coordinate_dim': 32,
                                           'resolution': [64, 64, 64, 50]}),
)


def merge_hparams(args, config):
    """utils/params_utils.py:1-8 of the reference."""
    for section in ("OptimizationParams", "ModelHiddenParams", "ModelParams", "PipelineParams"):
        for k, v in config.get(section, {}).items():
            if hasattr(args, k):
                setattr(args, k, v)
    return args


def default_args(time_resolution=None, **overrides):
    """Namespace holding every group's defaults with the default config overlay applied -- what train_4DGS.py
    sees when started without flags."""
    parser = ArgumentParser()
    groups = (ModelParams(parser), OptimizationParams(parser), PipelineParams(parser), ModelHiddenParams(parser))
    args = parser.parse_args([])
    args = merge_hparams(args, DEFAULT_CONFIG)
    if time_resolution is not None:
        kc = dict(args.kplanes_config)
        kc["resolution"] = list(kc["resolution"][:3]) + [int(time_resolution)]
        args.kplanes_config = kc
    for k, v in overrides.items():
        setattr(args, k, v)
    lp, op, pp, hp = (g.extract(args) for g in groups)
    return args, lp, op, pp, hp


def get_combined_args(parser: ArgumentParser):
    cmdlne_string = sys.argv[1:]
    cfgfile_string = "Namespace()"
    args_cmdline = parser.parse_args(cmdlne_string)
    try:
        cfgfilepath = os.path.join(args_cmdline.input_dir, "cfg_args")
        print("Looking for config file in", cfgfilepath)
        with open(cfgfilepath) as cfg_file:
            print("Config file found: {}".format(cfgfilepath))
            cfgfile_string = cfg_file.read()
    except TypeError:
        print("Config file not found at")
    args_cfgfile = eval(cfgfile_string)
    merged = vars(args_cfgfile).copy()
    for k, v in vars(args_cmdline).items():
        if v is not None:
            merged[k] = v
    return Namespace(**merged)
