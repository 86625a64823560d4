"""Parameter groups with the reference's defaults (reference arguments/__init__.py:19-174) and the default
config overlay arguments/dnerf/hellwarrior.py -> dnerf_default.py (applied by `default_config()` because mmcv,
which the reference uses to read those files, is not a dependency here)."""
import os
import sys
from argparse import ArgumentParser, Namespace


class GroupParams:
    """Plain attribute bag returned by ParamGroup.extract()."""


# Defaults of the four parameter groups, as data: option name -> default value (reference arguments/__init__.py:47-174).
# A leading underscore marks the options that also get a one-letter flag (-s, -m, -i, -r, -w), as in the reference.
_MODEL = {
    "sh_degree": 3, "_source_path": "", "_model_path": "", "_images": "images", "_resolution": -1, "_white_background": False,
    "data_device": "cuda", "eval": True, "render_process": False, "add_points": False, "extension": ".png", "llffhold": 8,
}
_PIPELINE = {"convert_SHs_python": False, "compute_cov3D_python": False, "debug": False}
_HIDDEN = {
    "net_width": 64, "timebase_pe": 4, "defor_depth": 1, "posebase_pe": 10, "scale_rotation_pe": 2, "opacity_pe": 2,
    "timenet_width": 64, "timenet_output": 32, "bounds": 1.6,
    "plane_tv_weight": 0.0001, "time_smoothness_weight": 0.01, "l1_time_planes": 0.0001,
    "kplanes_config": {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]},
    "multires": [1, 2, 4, 8],
    "no_dx": False, "no_grid": False, "no_ds": False, "no_dr": False, "no_do": True, "no_dshs": True,
    "empty_voxel": False, "grid_pe": 0, "static_mlp": False, "apply_rotation": False,
}
_OPTIM = {
    "dataloader": False, "zerostamp_init": False, "custom_sampler": None,
    "iterations": 30_000, "coarse_iterations": 3000,
    "position_lr_init": 0.00016, "position_lr_final": 0.0000016, "position_lr_delay_mult": 0.01, "position_lr_max_steps": 20_000,
    "deformation_lr_init": 0.00016, "deformation_lr_final": 0.000016, "deformation_lr_delay_mult": 0.01,
    "grid_lr_init": 0.0016, "grid_lr_final": 0.00016,
    "feature_lr": 0.0025, "opacity_lr": 0.05, "scaling_lr": 0.005, "rotation_lr": 0.001,
    "percent_dense": 0.01, "lambda_dssim": 0, "lambda_lpips": 0,
    "weight_constraint_init": 1, "weight_constraint_after": 0.2, "weight_decay_iteration": 5000,
    "opacity_reset_interval": 3000, "densification_interval": 100, "densify_from_iter": 500, "densify_until_iter": 15_000,
    "densify_grad_threshold_coarse": 0.0002, "densify_grad_threshold_fine_init": 0.0002, "densify_grad_threshold_after": 0.0002,
    "pruning_from_iter": 500, "pruning_interval": 100,
    "opacity_threshold_coarse": 0.005, "opacity_threshold_fine_init": 0.005, "opacity_threshold_fine_after": 0.005,
    "batch_size": 1, "add_point": False,
}


class ParamGroup:
    """One argparse group built from a table of defaults.  Booleans become store_true flags, everything else takes the type
    of its default; with `fill_none` every default is None (the render scripts use that to tell "not given" apart)."""
    TITLE, TABLE = "", {}

    def __init__(self, parser: ArgumentParser, fill_none=False):
        self._names = set()
        group = parser.add_argument_group(self.TITLE)
        for raw, default in self.TABLE.items():
            name = raw.lstrip("_")
            self._names.add(name)
            setattr(self, raw, default)                      # the defaults stay readable as attributes
            flags = ["--" + name] + (["-" + name[0]] if raw.startswith("_") else [])
            shown = None if fill_none else default
            if isinstance(default, bool):
                group.add_argument(*flags, default=shown, action="store_true")
            else:
                group.add_argument(*flags, default=shown, type=type(default))

    def extract(self, args):
        out = GroupParams()
        for name, value in vars(args).items():
            if name in self._names:
                setattr(out, name, value)
        return out


class ModelParams(ParamGroup):
    TITLE, TABLE = "Loading Parameters", _MODEL

    def __init__(self, parser, sentinel=False):
        super().__init__(parser, fill_none=sentinel)

    def extract(self, args):
        out = super().extract(args)
        out.source_path = os.path.abspath(out.source_path)
        return out


class PipelineParams(ParamGroup):
    TITLE, TABLE = "Pipeline Parameters", _PIPELINE


class ModelHiddenParams(ParamGroup):
    TITLE, TABLE = "ModelHiddenParams", _HIDDEN


class OptimizationParams(ParamGroup):
    TITLE, TABLE = "Optimization Parameters", _OPTIM


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
    """Command line merged over the `cfg_args` file a training run left in its model directory (a printed Namespace): what
    the render scripts start from.  A command-line value wins whenever it was actually given (is not None)."""
    given = parser.parse_args(sys.argv[1:])
    stored = Namespace()
    try:
        path = os.path.join(given.input_dir, "cfg_args")
        print("Looking for config file in", path)
        with open(path) as fh:
            text = fh.read()
        print("Config file found: {}".format(path))
        stored = eval(text)                                   # the file holds repr(Namespace(...)), as the reference writes it
    except TypeError:
        print("Config file not found at")
    merged = dict(vars(stored))
    merged.update({k: v for k, v in vars(given).items() if v is not None})
    return Namespace(**merged)
