"""Child process of tests/test_gpu_pipeline.py::test_rccl_branch_world_size_1: runs every collective of the path on the
"nccl" (= RCCL) backend with ONE rank on the test GPU and prints RCCL_WORLD1_OK.  A fresh process: the process group is
created before anything else touches the GPU, nothing is re-exec'ed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as td
    from ipdm_pytorch_amd import dist as idist, synth
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1)
    assert idist.describe() == {"world": 1, "backend": "nccl"}
    dev = "cuda:0"
    # 1. the path's one collective, not short-circuited: [b,1,512,512] f32 block in, the same block out, bit for bit
    x = torch.from_numpy(synth.hash_normal((3, 1, 512, 512), 11)).to(dev)
    y = idist.all_gather_slices(x, 3, 0, 1, force_collective=True)
    assert y.data_ptr() != x.data_ptr() and y.is_cuda and torch.equal(x, y), "all_gather_into_tensor over RCCL changed the data"
    # 2. the timing reduction on a device tensor (MAX all-reduce, float64)
    v = idist.max_over_ranks(1.2345678901234567, dev)
    assert v == 1.2345678901234567, v
    # 3. the adaptive-schedule branch reduction of the denoiser (denoiser._rank_max)
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
    stub = type("S", (), {"proj_device": dev, "force_collectives": True})()
    fn = progressive_domain_denoiser._rank_max(stub)
    assert fn is not None and fn(0.3125) == 0.3125
    idist.barrier()
    torch.cuda.synchronize()
    td.destroy_process_group()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
