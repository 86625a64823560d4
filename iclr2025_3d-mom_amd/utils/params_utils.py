"""utils/params_utils.py of the reference: overlay a config file's sections on the parsed arguments."""


def merge_hparams(args, config):
    for section in ("OptimizationParams", "ModelHiddenParams", "ModelParams", "PipelineParams"):
        if section in config.keys():
            for key, value in config[section].items():
                if hasattr(args, key):
                    setattr(args, key, value)
    return args
