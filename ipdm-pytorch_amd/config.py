"""Option handling of the sampling path: mirror of Config/default_config.py (argparse flags, JSON
overlay, cfg_load).  Only the keys the hot path consumes plus the bookkeeping keys
the harness prints are defined; names, types and defaults are the reference's."""
import argparse
import json
import sys

# (name, type, default, nargs)  -- Config/default_config.py:10-157
_FLAGS = [
    ("test_batch_size", int, 1, None), ("mode", str, "train_img", None), ("run_name", str, "default", None),
    ("model_name", str, "IPDM", None), ("device", str, "cuda:0", None), ("convertor", str, "TV", None),
    ("load_option_path", str, None, None), ("load_img_model_path", str, None, None),
    ("load_proj_model_path", str, None, None), ("resume_epochs_proj", int, 0, None),
    ("resume_epochs_img", int, 0, None), ("display_result", bool, False, None),
    ("test_result_data_save", bool, False, None), ("benchmark_test", bool, False, None),
    ("metrics", str, ["psnr", "ssim", "fsim", "vif", "nqm"], "+"), ("fbp_sharpen", bool, False, None),
    ("ntv", int, 0, None), ("normal", bool, False, None), ("ultra_img_denoise", bool, True, None),
    # img model
    ("in_channels_img", int, 1, None), ("out_channels_img", int, 1, None), ("model_channels_img", int, 64, None),
    ("attention_resolutions_img", int, [16], "+"), ("channel_mult_img", float, [1, 1, 2, 2, 4, 4], "+"),
    ("timesteps_img", int, 1000, None), ("partial_timesteps_img", int, 50, None), ("schedule_power_img", float, 1, None),
    ("clip_img", bool, True, None), ("save_states_img", bool, False, None), ("lambda_ratio_img", float, 5, None),
    ("t_start_img", int, None, "+"), ("eta_img", float, 0.5, None), ("constant_guidance_img", float, None, None),
    ("kernel_size_img", int, 4, None), ("amplitude_img", float, 20, None), ("ddim_timesteps_img", int, [1, 2, 2], "+"),
    ("sample_method_img", str, "dense", None), ("save_it_state_img", bool, False, None),
    # proj model
    ("in_channels_proj", int, 1, None), ("out_channels_proj", int, 1, None), ("model_channels_proj", int, 64, None),
    ("attention_resolutions_proj", int, [32], "+"),
    ("channel_mult_proj", float, [1 / 64, 2 / 64, 4 / 64, 2, 2, 4, 4], "+"), ("timesteps_proj", int, 1000, None),
    ("partial_timesteps_proj", int, 50, None), ("schedule_power_proj", float, 1, None), ("clip_proj", bool, False, None),
    ("lambda_ratio_proj", float, 5, None), ("t_start_proj", int, None, "+"), ("eta_proj", float, 0.4, None),
    ("constant_guidance_proj", float, None, None), ("kernel_size_proj", int, 4, None),
    ("amplitude_proj", float, 5, None), ("ddim_timesteps_proj", int, [1, 2, 2], "+"),
    ("sample_method_proj", str, "dense", None), ("save_it_state_proj", bool, False, None),
    ("dose", float, 0.25, None),
    # whole-dataset evaluation (test()/fit(), Config/default_config.py:19,141-149)
    ("test_numbers", int, 50, None), ("data_type", str, "siemens", None),
    ("test_dataset_path_FD_img", str, None, None), ("test_dataset_path_LD_img", str, None, None),
    ("test_dataset_path_FD_proj", str, None, None), ("test_dataset_path_LD_proj", str, None, None),
]


def default_cfg(argv=None):
    """Config/default_config.py:7-172: argparse defaults, then --load_option_path JSON overlays every
    key that was not given on the command line."""
    parser = argparse.ArgumentParser("IPDM sampling-path options")
    for name, typ, default, nargs in _FLAGS:
        kw = dict(type=typ, default=default)
        if nargs:
            kw["nargs"] = nargs
        parser.add_argument("--" + name, **kw)
    argv = sys.argv[1:] if argv is None else argv
    opt = parser.parse_args(argv)
    given = [a[2:] for a in argv if "--" in a]
    if opt.load_option_path is not None:
        load_option(opt, opt.load_option_path, given)
    return opt


def cfg_load(new_cfg, old_cfg):
    """Config/default_config.py:176-185: overwrite EXISTING keys only; unknown keys warn and are ignored."""
    for key in new_cfg.keys():
        if isinstance(new_cfg[key], dict):
            cfg_load(new_cfg[key], old_cfg[key])
        elif key in old_cfg.keys():
            old_cfg[key] = new_cfg[key]
        else:
            print(f"no key names {key} in config\n")


def load_option(opt, load_path, exception):
    """Config/default_config.py:188-194."""
    with open(load_path, "r") as f:
        loaded = json.load(f)
    for key in exception:
        loaded.pop(key, None)
    cfg_load(loaded, opt.__dict__)


def mayo_test_options():
    """The values of Config/Mayo-Config/test_progressive_option.json that the hot path reads, with
    convertor="FBP" (north_star) -- used when no JSON file is at hand (bench, smoke)."""
    return dict(mode="test_prog", convertor="FBP", fbp_sharpen=True, normal=False, ultra_img_denoise=True,
                attention_resolutions_img=[8, 16], channel_mult_img=[1, 1, 2, 2, 4, 4], schedule_power_img=1,
                clip_img=True, lambda_ratio_img=10, t_start_img=[15, 15, 15], eta_img=0.7, constant_guidance_img=0.45,
                kernel_size_img=4, amplitude_img=30,
                attention_resolutions_proj=[16, 32], channel_mult_proj=[0.0625, 0.125, 0.25, 2, 2, 4, 4],
                schedule_power_proj=5, clip_proj=False, lambda_ratio_proj=1, t_start_proj=[15, 15, 15], eta_proj=0.5,
                constant_guidance_proj=None, kernel_size_proj=4, amplitude_proj=7)
