"""The multi-device runner below Python (snn_pool, include/snn_c.h; shadernn_amd/host/pool.cpp) and the snn_run CLI.  On the one-GPU test box the
replicas are two contexts (host thread + stream each) on device 0: the split / gather logic and the threading are what is under test; the same
code gives every GPU of a node its own replica."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model_file(tmp_path, net, W, H, name):
    from shadernn_amd import models

    return models.write_json(net, W, H, str(tmp_path / name), bin_weights=True)


def test_c3_as_two_in_process_replicas_equals_the_single_context_run(ctx, tmp_path):
    """BASELINE configs[2]'s net (ResNet-18 224x224 fp32) as a global batch of 8 over two replicas: every image is computed exactly once, by the
    replica the shard rule names, and equals -- bit for bit -- what a single context computes for the same micro-batch."""
    from shadernn_amd import host, models

    net = models.resnet18(seed=1)
    H = W = 224
    path = _model_file(tmp_path, net, W, H, "resnet18.json")
    x = np.random.default_rng(21).random((8, H, W, 3), dtype=np.float32)
    pool = host.Pool(path, W, H, 3, devices=[0, 0], global_batch=8)
    assert pool.replicas() == 2 and pool.shard(0) == (0, 4, 1) and pool.shard(1) == (4, 4, 1)
    pool.upload(x)
    sec = pool.run(steps=1)
    assert sec > 0
    y = pool.output()
    assert y.shape[0] == 8 and np.isfinite(y).all()
    single = host.Model(path, W, H, 3, device=0, capture_graph=True, batch=4)
    for g in range(2):
        np.testing.assert_array_equal(y[4 * g : 4 * g + 4].reshape(4, -1), np.asarray(single(x[4 * g : 4 * g + 4])).reshape(4, -1), err_msg="replica %d" % g)
    single.close()
    # and against the oracle, image 5 (computed by replica 1)
    want = O.forward(net, x[5:6], threads=os.cpu_count() or 1)
    np.testing.assert_allclose(y[5].reshape(-1), want.reshape(-1), rtol=1e-4, atol=1e-4)
    # several steps in flight per replica, results unchanged (the timed mode of bench.py / snn_run --devices)
    pool.run(steps=3)
    np.testing.assert_array_equal(pool.output(), y)
    pool.close()


def test_pool_micro_batches_uneven_shares_and_the_rccl_edge_gather(ctx, tmp_path):
    """5 images over 2 replicas in micro-batches of 2: shares 2 + 3, slots [2] and [2, 1]; the gather puts every image where it belongs.  The RCCL
    gather refuses two ranks on one device (-3) and works on a one-replica pool (a one-rank communicator: the collective path itself runs)."""
    from shadernn_amd import host, models

    net = models.espcn_weights(seed=1)
    H, W = 40, 56
    path = _model_file(tmp_path, net, W, H, "espcn.json")
    x = np.random.default_rng(22).random((5, H, W, 1), dtype=np.float32)
    want = O.forward(net, x)
    pool = host.Pool(path, W, H, 1, devices=[0, 0], global_batch=5, micro_batch=2)
    assert pool.shard(0) == (0, 2, 1) and pool.shard(1) == (2, 3, 2)
    pool.upload(x)
    pool.run()
    np.testing.assert_allclose(pool.output(), want, rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="-3"):
        pool.output(rccl=True)
    pool.close()
    one = host.Pool(path, W, H, 1, devices=[0], global_batch=4, micro_batch=2)
    one.upload(x[:4])
    one.run()
    y_host = one.output()
    np.testing.assert_allclose(y_host, want[:4], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(one.output(rccl=True), y_host)
    one.close()


def _splitmix_unit_floats(n, seed=7767517):
    """tools/snn_run.cpp's input generator"""
    out = np.empty(n, np.float32)
    s = seed
    M = (1 << 64) - 1
    for i in range(n):
        s = (s + 0x9E3779B97F4A7C15) & M
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out[i] = np.float32(z >> 40) * np.float32(1.0 / 16777216.0)
    return out


def test_snn_run_cli_prints_the_layer_table_and_the_right_output(ctx, tmp_path):
    """lib/snn_run (the role of the reference's inferenceProcessorTest): loads a JSON model, runs it, prints the per-layer mean / population sigma
    table (first 5 loops dropped) and an output checksum that matches the oracle on the CLI's own synthetic input; --devices runs it through snn_pool."""
    from shadernn_amd import models

    net = models.espcn_weights(seed=1)
    H, W = 24, 32
    path = _model_file(tmp_path, net, W, H, "espcn.json")
    cli = os.path.join(ROOT, "shadernn_amd", "lib", "snn_run")
    assert os.path.exists(cli), "build() did not produce lib/snn_run"
    x = _splitmix_unit_floats(2 * H * W).reshape(2, H, W, 1)
    want = float(O.forward(net, x).astype(np.float64).sum())
    r = subprocess.run([cli, path, "--w", str(W), "--h", str(H), "--c", "1", "--batch", "2", "--loops", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().split("\n")
    assert lines[0].startswith("id") and "mean ms" in lines[0] and any("Conv2D" in l for l in lines) and any("Total GPU runtime" in l for l in lines), r.stdout
    got = float(lines[-1].split("checksum")[1].split()[0])
    assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (got, want)
    r2 = subprocess.run([cli, path, "--w", str(W), "--h", str(H), "--c", "1", "--batch", "2", "--loops", "3", "--devices", "0,0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        text=True, timeout=300)
    assert r2.returncode == 0 and r2.stdout.startswith("replicas 2"), r2.stdout + r2.stderr[-2000:]
    got2 = float(r2.stdout.split("checksum")[1].split()[0])
    assert abs(got2 - want) <= 1e-3 * max(1.0, abs(want)), (got2, want)
