"""Data-parallel path on CPU: world_size 2 over gloo.  The only exchange step of the hot path is one
all-reduce of the flat gradient buffer per step; the trainer shards the data by rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from probav_amd.trainClass import ModelTrainer, allreduce_mean_
    from tests.test_trainer_host import _stub, _l1, _metric, cpu_optimizer as make_optimizer
    try:
        # 1. the collective itself: mean over ranks of the flat 535 267-element buffer
        g = torch.full((535267,), float(rank + 1))
        allreduce_mean_(g)
        assert torch.allclose(g, torch.full_like(g, 1.5))
        # 2. two ranks, different shards, identical weights after every step
        torch.manual_seed(0)
        model = _stub()
        tr = ModelTrainer(model, _l1, _metric, make_optimizer("sgd", model, 0.1), os.path.join(tmp, "ck"), os.path.join(tmp, "lg"), multiGPU=True)
        n = 16
        rng = np.random.default_rng(5)
        X = np.zeros((n, 22, 22, 9, 1), np.float32)
        y = rng.normal(size=(n, 48, 48, 1)).astype(np.float32) * (1 + np.arange(n).reshape(n, 1, 1, 1))
        m = np.ones((n, 48, 48, 1), bool)
        tr.fitTrainData(X, [y, m], 4, 2, [X, y, m], saveBestOnly=False)
        assert tr.step == 4                          # 8 samples per rank / 4 * 2 epochs
        mine = model.flat.detach().clone()
        other = mine.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(mine, other), "replicas diverged"
        # 3. a data set that does not divide by the world size (17 samples -> 9 / 8 per rank used to give the ranks different
        #    batch counts and hang the tail all-reduce): shards are cut to len // world, both ranks take the same number of steps
        model3 = _stub()
        tr3 = ModelTrainer(model3, _l1, _metric, make_optimizer("sgd", model3, 0.1), os.path.join(tmp, "ck3"), os.path.join(tmp, "lg3"), multiGPU=True, evalStep=2)
        n3 = 17
        X3 = np.zeros((n3, 22, 22, 9, 1), np.float32)
        y3 = rng.normal(size=(n3, 48, 48, 1)).astype(np.float32)
        m3 = np.ones((n3, 48, 48, 1), bool)
        tr3.fitTrainData(X3, [y3, m3], 3, 3, [X3[:6], y3[:6], m3[:6]], valSteps=1, saveBestOnly=False)
        assert tr3.step == 8                         # 8 samples per rank * 3 epochs / batch 3 = 8 batches (the last one partial)
        steps = torch.tensor([float(tr3.step)])
        dist.all_reduce(steps)
        assert float(steps) == 16.0
        # 4. C2: logged loss / metric are means over the replicas -> identical running means on both ranks
        pair = torch.tensor([tr3.testLoss.result(), tr3.testPSNR.result(), tr3.trainLoss.result(), tr3.trainPSNR.result()], dtype=torch.float64)
        other = pair.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(pair, other), "scalar metrics are not reduced over the replicas"
        # 5. checkpoints under DP: only rank 0 writes (the replicas are identical), and what it wrote restores on any rank
        dist.barrier()
        names = open(os.path.join(tmp, "ck3", "checkpoint.pt-index")).read().split()
        assert names and all(n.endswith(".pt") for n in names) and len(names) <= 5
        model4 = _stub()
        tr4 = ModelTrainer(model4, _l1, _metric, make_optimizer("sgd", model4, 0.1), os.path.join(tmp, "ck3"), os.path.join(tmp, "lg4_%d" % rank), multiGPU=True)
        assert tr4.step > 0 and tr4.save_counter == int(names[-1][5:-3])
        if rank == 0:
            open(os.path.join(tmp, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_replica_consistency(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
