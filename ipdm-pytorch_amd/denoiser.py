"""progressive_domain_denoiser: drop-in for the sampling-path surface of the reference's harness
class (Utils/train_test_utils.py:121-594): update_opt / reset_opt / temp_clear / data_sample_load /
proj_denoiser / img_denoiser / progressive_denoiser and the result dictionaries.

Everything between data_sample_load and the returned tensor stays on the GPU; slices of a batch are
independent (per-slice semantics) and may be sharded over ranks (`world=` keyword, see dist.py).
The SURVEY 8(f) rows are mixed in from their own modules: the ART convertor (art.py), dataset iteration / metric and
result files / test() / fit() (evaluate.py), opt.normal (normalize.py).  Training and figure rendering are out of scope
and raise NotImplementedError when asked for.
"""
import copy
import os

import numpy as np
import torch

from . import _lib
from .config import cfg_load
from .diffusion import GaussianDiffusion, NoiseSource
from .fbp import FBP, tensor_sharpen
from .evaluate import EvaluationMixin
from .normalize import yeo_johnson_transform
from .unet import UNetModel


class DotDict(dict):
    def __setattr__(self, key, value):
        self[key] = value

    def __getattr__(self, key):
        value = self[key]
        return DotDict(value) if isinstance(value, dict) else value


class ResultTempDict(DotDict):
    """Utils/train_test_utils.py:45-56: int index k>0 -> "iter_k", -1 -> last."""

    def __getitem__(self, item):
        if isinstance(item, str):
            return super().__getitem__(item)
        if isinstance(item, int):
            if item > 0:
                return self[f"iter_{item}"]
            if item == -1:
                return self[f"iter_{len(self)}"]


def miu2pixel(miu):
    """Dataset/npz_data_loader.py:20-36 (numpy / torch, as the reference's)."""
    hu = (miu - 0.183) * 1e3 / 0.183 - 24
    img = (hu + 1024) / 4096
    img = img.clone() if isinstance(img, torch.Tensor) else np.array(img, copy=True)
    img[hu < -1024] = 0
    img[hu > 3072] = 1
    return img


class progressive_domain_denoiser(EvaluationMixin):
    def __init__(self, opt, result_save_path=None, seed=0, slice_id0=0):
        self.opt = opt
        self.opt_temp = copy.deepcopy(opt)
        self.rank = torch.distributed.get_rank() if torch.distributed.is_available() and \
            torch.distributed.is_initialized() else 0
        self.seed, self.slice_id0 = seed, slice_id0
        self.result_save_path = result_save_path
        self.proj_model = self.img_model = None
        if opt.mode in ("train_proj", "train_img"):
            raise NotImplementedError("training is out of scope of the sampling hot path")
        if opt.mode in ("test_proj", "test_prog"):
            self.init_proj_model()
        self.init_convertor(opt.convertor)
        if opt.mode in ("test_img", "test_prog"):
            self.init_img_model()
        self.load_model()
        self.fdct = self.fdproj = self.ldct = self.ldct_np = self.ldproj = self.ldproj_np = None
        self.proj_denoise_result = ResultTempDict()
        self.proj_denoise_convert2img_result = ResultTempDict()
        self.img_denoise_result = ResultTempDict()
        self.progressive_denoise_result = ResultTempDict()
        self.noise_strength = None
        self.trans_ldproj = self.trans_ldimg = None     # fitted power transforms of the loaded sample (opt.normal)
        self.noise = None          # optional NoiseSource / InjectedNoise override (parity tests)
        # metric bookkeeping and the result tree <result_save_path>/<model>_<run>/save_test_results (:131-133,156-200);
        # without a result_save_path nothing is created until test() / save_path_load() is asked for
        self._init_evaluation(None if result_save_path is None else
                              os.path.join(result_save_path, "%s_%s" % (opt.model_name, opt.run_name)))

    # ------------------------------------------------------------------ options (:202-211)
    def update_opt(self, ultra_cfg=None):
        if ultra_cfg is not None:
            cfg_load(ultra_cfg, self.opt.__dict__)
        if "convertor" in ultra_cfg.keys():          # update_opt(None) raises AttributeError as the reference does
            self.init_convertor(ultra_cfg["convertor"])

    def reset_opt(self):
        self.opt = copy.deepcopy(self.opt_temp)

    # ------------------------------------------------------------------ models (:213-251)
    def init_img_model(self):
        o = self.opt
        self.img_model = UNetModel(in_channels=o.in_channels_img, model_channels=o.model_channels_img,
                                   out_channels=o.out_channels_img, attention_resolutions=o.attention_resolutions_img,
                                   channel_mult=o.channel_mult_img).to(o.device)
        self.img_device = torch.device(o.device)
        self.img_dtype = torch.float32
        self.img_gaussian_diffusion = GaussianDiffusion(timesteps=o.timesteps_img, beta_schedule="cosine",
                                                        schedule_power=o.schedule_power_img)

    def init_proj_model(self):
        o = self.opt
        self.proj_model = UNetModel(in_channels=o.in_channels_proj, model_channels=o.model_channels_proj,
                                    out_channels=o.out_channels_proj,
                                    attention_resolutions=o.attention_resolutions_proj,
                                    channel_mult=o.channel_mult_proj).to(o.device)
        self.proj_device = o.device
        self.proj_dtype = torch.float32
        self.proj_gaussian_diffusion = GaussianDiffusion(timesteps=o.timesteps_proj, beta_schedule="cosine",
                                                         schedule_power=o.schedule_power_proj)

    def init_convertor(self, convertor):
        """Utils/train_test_utils.py:225-233.  The reference reads Recon/Simens_alut.txt / Simens_theta.txt from its
        working directory; here the two tables are regenerated (art.area_lut / art.view_angles reproduce the files)."""
        from . import art
        self._fbp = None
        self._art = None
        if convertor == "FBP":
            self._fbp = FBP(device=self.opt.device, **(getattr(self, "fbp_geometry", None) or {}))
            self.convertor = self._fbp.convert
        elif convertor == "ART":
            self._art = art._plan_for(*self._art_tables(), torch.device(self.opt.device))
            self.convertor = lambda x: art.recons_torch(x, *self._art_tables(), nstart=10, ntv=self.opt.ntv,
                                                        sample_rate=1, permute=True, device=self.opt.device)
        else:
            self.convertor = None
        self.projection = lambda x: art.proj_torch(x, *self._art_tables(), device=self.opt.device)

    def set_fbp_geometry(self, **geometry):
        """Extension (not in the reference, whose FBP geometry is hard-coded): plan the FBP convertor for another
        sinogram shape, e.g. fbp.ALT_GEOMETRY (1152 views x 736 detectors); no arguments = the reference geometry."""
        self.fbp_geometry = dict(geometry) if geometry else None
        if self.opt.convertor == "FBP":
            self.init_convertor("FBP")

    def _art_tables(self):
        from . import art
        if getattr(self, "_art_tab", None) is None:
            self._art_tab = (art.area_lut(), art.view_angles())
        return self._art_tab

    def load_model(self):
        """Utils/train_test_utils.py:247-251 + LoggerX.load_checkpoints / load_network (Utils/loggerx.py:69-80,131-140):
        state_dict files `<load path>/{proj_model,img_model}-<epoch>` (what LoggerX.checkpoints writes into its
        save_models directory), `module.` removed from the keys.  A file that does not exist is skipped as in the
        reference (osp.exists guard) -- but not silently: the network then keeps its initial weights."""
        import warnings
        o = self.opt
        for name, model, ep, path in (("img_model", self.img_model, o.resume_epochs_img, o.load_img_model_path),
                                      ("proj_model", self.proj_model, o.resume_epochs_proj, o.load_proj_model_path)):
            if ep > 0 and path is not None and model is not None:
                f = os.path.join(path, "%s-%d" % (name, ep))
                if not os.path.exists(f) and os.path.exists(os.path.join(path, "save_models", "%s-%d" % (name, ep))):
                    f = os.path.join(path, "save_models", "%s-%d" % (name, ep))      # a run directory was given
                if not os.path.exists(f):
                    warnings.warn("checkpoint %s not found: %s keeps its initial weights" % (f, name), UserWarning)
                    continue
                sd = torch.load(f, map_location="cpu")
                model.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})

    # ------------------------------------------------------------------ temp (:397-419)
    def temp_clear(self):
        self.proj_temp_clear()
        self.img_temp_clear()
        self.metric_clear()
        self.noise_strength = None

    def proj_temp_clear(self):
        self.proj_denoise_convert2img_result = ResultTempDict()
        self.proj_denoise_result = ResultTempDict()

    def img_temp_clear(self):
        self.img_denoise_result = ResultTempDict()
        self.progressive_denoise_result = ResultTempDict()

    def _noise(self):
        if self.noise is None:
            self.noise = NoiseSource(self.seed, self.slice_id0)
        return self.noise

    # ------------------------------------------------------------------ data (:569-594)
    def data_sample_load(self, ldct=None, ldproj=None, fdproj=None, fdct=None):
        if ldct is not None:
            if self.opt.normal:        # :578-580
                ldct_norm, self.trans_ldimg = yeo_johnson_transform(ldct)
                self.ldct = ldct_norm.to(self.opt.device)
            else:
                self.ldct = ldct.to(self.opt.device)
            self.ldct_np = miu2pixel(ldct.squeeze().cpu().numpy())
        if ldproj is not None:
            if self.opt.normal:        # :585-587
                ldproj_norm, self.trans_ldproj = yeo_johnson_transform(ldproj)
                self.ldproj = ldproj_norm.to(self.opt.device)
            else:
                self.ldproj = ldproj.to(self.opt.device)
            self.ldproj_np = ldproj.squeeze().cpu().numpy()
        if fdct is not None:
            self.fdct = miu2pixel(fdct).squeeze().numpy()
        if fdproj is not None:
            self.fdproj = fdproj.squeeze().numpy()

    # ------------------------------------------------------------------ device-resident core
    def _convert_dev(self, sino_b1hw, gain):
        if self._fbp is not None:
            return self._fbp.convert_device(sino_b1hw[:, 0], flip=True, gain=gain).unsqueeze(1)
        if self._art is not None:       # recons_torch(G * x, nstart=10, ntv, permute=True), :231-232
            x = sino_b1hw[:, 0] if gain == 1 else sino_b1hw[:, 0] * float(gain)
            return self._art.reconstruct_device(x, 10, self.opt.ntv).permute(0, 2, 1).contiguous().unsqueeze(1)
        raise NotImplementedError("convertor %r: only 'FBP' and 'ART' exist" % (self.opt.convertor,))

    def _proj_dense(self, x):
        o = self.opt
        if o.sample_method_proj == "sparse":       # Utils/train_test_utils.py:445-453
            res = self.proj_gaussian_diffusion.sparse_guided_reverse_process(
                model=self.proj_model, condition=x.to(self.proj_device, torch.float32), t_start=o.t_start_proj,
                condition_lambda_max=0.49, condition_lambda_min=0.35, clip_denoised=o.clip_proj,
                ddim_timesteps=o.ddim_timesteps_proj, eta=o.eta_proj, noise=self._noise())
            return res, None, self.noise_strength
        if o.sample_method_proj != "dense":
            raise ValueError("sample_method_proj must be 'dense' or 'sparse'")
        return self.proj_gaussian_diffusion.guided_reverse_process(
            model=self.proj_model, img=x.to(self.proj_device, torch.float32), t_start=o.t_start_proj, clip=o.clip_proj,
            lambda_ratio=o.lambda_ratio_proj, eta=o.eta_proj, mode="proj", constant_guidance=o.constant_guidance_proj,
            kernel_size_proj=o.kernel_size_proj, amplitude_proj=o.amplitude_proj, only_convertor=o.benchmark_test,
            normal=o.normal, transformer=self.trans_ldproj, noise=self._noise(), rank_max=self._rank_max())

    def _rank_max(self):
        """Adaptive pass schedule (t_start_proj=None) under slice sharding: the branch is taken on the maximum over
        ALL ranks' slices, as the reference takes it over its whole batch (Model/model.py:596-609)."""
        import torch.distributed as td
        if not (td.is_available() and td.is_initialized()
                and (td.get_world_size() > 1 or getattr(self, "force_collectives", False))):
            return None
        from . import dist as idist
        return lambda v: idist.max_over_ranks(v, self.proj_device)

    def _img_dense(self, x, noise_strength, ultra):
        o = self.opt
        if o.sample_method_img not in ("dense", "sparse"):
            raise ValueError("sample_method_img must be 'dense' or 'sparse'")
        xd = x.to(self.img_device, torch.float32).contiguous()
        common = dict(model=self.img_model, clip=o.clip_img, lambda_ratio=o.lambda_ratio_img,
                      save_states=o.save_states_img, noise_strength=noise_strength, ldct=xd, mode="img",
                      kernel_size_img=o.kernel_size_img, amplitude_img=o.amplitude_img,
                      only_convertor=o.benchmark_test, normal=o.normal, transformer=self.trans_ldimg, noise=self._noise())
        if o.sample_method_img == "sparse":        # Utils/train_test_utils.py:505-514
            result = self.img_gaussian_diffusion.sparse_guided_reverse_process(
                model=self.img_model, condition=xd, t_start=o.t_start_img, condition_lambda_max=0.5, condition_lambda_min=0.3,
                clip_denoised=True, ddim_timesteps=o.ddim_timesteps_img, eta=o.eta_img, noise=self._noise())
        else:
            result, _, _ = self.img_gaussian_diffusion.guided_reverse_process(
                img=xd, t_start=o.t_start_img, eta=o.eta_img, constant_guidance=o.constant_guidance_img, **common)
        if ultra:       # Utils/train_test_utils.py:515-536
            result_, _, _ = self.img_gaussian_diffusion.guided_reverse_process(
                img=result[-1], t_start=[5, 5, 5], eta=0.6, constant_guidance=0.6, **common)
            result = result + result_
        return result

    # ------------------------------------------------------------------ public path (:421-567)
    def proj_denoiser(self, x, convert=True, save_state=True, save_proj_state=False, return_idx=-1):
        result, _, noise_strength = self._proj_dense(x)
        self.noise_strength = noise_strength
        self.proj_temp_clear()
        G = 10 if self.opt.clip_proj else 1
        if save_proj_state:
            for it in range(len(result)):
                self.proj_denoise_result[f"iter_{it + 1}"] = result[it].cpu().numpy()
        if save_state:
            if convert:
                for it in range(len(result)):
                    self.proj_denoise_convert2img_result[f"iter_{it + 1}"] = self._convert_dev(result[it], G).cpu().numpy()
                return torch.from_numpy(self.proj_denoise_convert2img_result[f"iter_{len(result)}"]), self.noise_strength
            for it in range(len(result)):
                self.proj_denoise_result[f"iter_{it + 1}"] = result[it].cpu().numpy()
            return result[return_idx], self.noise_strength
        if convert:
            img = self._convert_dev(result[return_idx], G)
            self._last_convert_dev = img
            self.proj_denoise_convert2img_result["iter_1"] = img.cpu().numpy()
            return torch.from_numpy(self.proj_denoise_convert2img_result["iter_1"]), self.noise_strength
        self.proj_denoise_result["iter_1"] = result[return_idx].cpu().numpy()
        return result[return_idx], self.noise_strength

    def img_denoiser(self, x, return_idx=-1, noise_strength=None, mode="progressive", sharpen_num=45, save_state=True):
        result = self._img_dense(x, noise_strength, self.opt.ultra_img_denoise)
        self.img_temp_clear()
        target = self.progressive_denoise_result if mode == "progressive" else self.img_denoise_result
        if save_state:
            for it in range(len(result)):
                target[f"iter_{it + 1}"] = result[it].cpu().numpy()
        else:
            target["iter_1"] = result[return_idx].cpu().numpy()
        return result[return_idx]

    def progressive_denoiser(self, save_proj_state=False, convert=True, sharpen_num=42):
        result, n_s = self.proj_denoiser(self.ldproj, save_state=self.opt.save_it_state_proj,
                                         save_proj_state=save_proj_state, convert=convert)
        if not (self.opt.convertor == "FBP" and self.opt.fbp_sharpen):
            sharpen_num = -1
        x = tensor_sharpen(result.to(self.opt.device), sharpen_num)
        if self.opt.normal:         # :560-562
            x, self.trans_ldimg = yeo_johnson_transform(x)
        return self.img_denoiser(x, noise_strength=n_s, save_state=self.opt.save_it_state_img)

    # ------------------------------------------------------------------ fast path (no host copies)
    @torch.no_grad()
    def proj_denoiser_device(self, ldproj=None):
        """proj_denoiser(convert=True, save_state=False) with nothing copied to the host: the projection-domain loop and
        the convertor.  Returns (image [B,1,G,G] on the device, noise_strength)."""
        x = self.ldproj if ldproj is None else ldproj
        result, _, n_s = self._proj_dense(x)
        self.noise_strength = n_s
        return self._convert_dev(result[-1], 10 if self.opt.clip_proj else 1), n_s

    @torch.no_grad()
    def progressive_denoiser_device(self, ldproj=None, sharpen_num=42):
        """Same arithmetic as progressive_denoiser(save_* = False) with every intermediate kept on the
        GPU and no result-dict copies: what bench.py times.  Returns the device tensor [B,1,512,512]."""
        img, n_s = self.proj_denoiser_device(ldproj)
        if self.opt.convertor == "FBP" and self.opt.fbp_sharpen:
            img = tensor_sharpen(img, sharpen_num)
        if self.opt.normal:
            img, self.trans_ldimg = yeo_johnson_transform(img)
        return self._img_dense(img, n_s, self.opt.ultra_img_denoise)[-1]


SMOKE_PROJ = dict(in_channels=1, model_channels=16, out_channels=1, attention_resolutions=[16],
                  channel_mult=[0.25, 0.25, 0.5, 1, 2, 4], num_heads=1)
SMOKE_IMG = dict(in_channels=1, model_channels=16, out_channels=1, attention_resolutions=[8],
                 channel_mult=[1, 1, 2, 2, 4], num_heads=1)


def smoke_pipeline(device):
    """Tiny end-to-end pass used by __graft_entry__.smoke(): reduced UNets (attention head dim 64 with
    one head), real FBP geometry, 2+2 proj steps, 2 img steps, ultra pass."""
    from . import synth
    from .config import default_cfg, mayo_test_options
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(device=device, t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=True), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=11)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(device)
    den.img_model = UNetModel(**SMOKE_IMG).to(device)
    sd_p = synth.synth_state_dict(den.proj_model._shapes, seed=21)
    sd_i = synth.synth_state_dict(den.img_model._shapes, seed=22)
    den.proj_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_p.items()})
    den.img_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_i.items()})
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(1)), seed=1)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    # record the noise the device generates so that the oracle replays the very same draws
    rec = _RecordingNoise(NoiseSource(11, 0))
    den.noise = rec
    out = den.progressive_denoiser(sharpen_num=70)
    inputs = dict(opt=copy.deepcopy(opt.__dict__), ldproj=sino, noise=[z.cpu() for z in rec.draws])
    return out.cpu().numpy(), inputs


class _RecordingNoise:
    def __init__(self, inner):
        self.inner, self.draws = inner, []

    def next_like(self, x):
        z = self.inner.next_like(x)
        self.draws.append(z)
        return z
