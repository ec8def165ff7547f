"""ONE process with an initialised RCCL world of one rank (`init_process_group("nccl", world_size=1)`) for the two hardware
bring-up checks that need it: the sequence-parallel exchange (test_sequence_parallel_gpu._nccl_world1_body) and the VAE's
point-to-point halos (test_vae_parallel_gpu._nccl_p2p_body).  RCCL start-up is 25-50 s on a fresh box: shared, not repeated."""
import os
import traceback

import torch
import torch.distributed as dist


def run(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from test_sequence_parallel_gpu import _nccl_world1_body
        from test_vae_parallel_gpu import _nccl_p2p_body
        for name, body in (("sp", _nccl_world1_body), ("vae", _nccl_p2p_body)):
            try:
                body(ret)
            except Exception:       # reported by the test that owns the check, the other one still runs
                ret[name + "_error"] = traceback.format_exc()
    finally:
        dist.destroy_process_group()
