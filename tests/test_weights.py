"""Weight formats either side of the path (viettts_amd/hifigan/weights.py)."""
import pickle

import numpy as np
import pytest

from viettts_amd.hifigan.config import TINY, TINY2, V1, HifiganConfig
from viettts_amd.hifigan.synth import synthetic_params
from viettts_amd.hifigan.weights import (check_params, conv_specs, haiku_to_state_dict, load_haiku_pickle, num_parameters,
                                          save_haiku_pickle, state_dict_to_haiku, torch_weight_to_haiku)


def test_inventory_matches_reference():
    specs = conv_specs(V1)
    assert len(specs) == 78  # 1 + 4 + 72 + 1
    assert num_parameters(V1) == 13926017
    keys = [s.key for s in specs]
    assert keys[0] == "generator/~/conv1_d" and keys[-1] == "generator/~/conv1_d_1"
    assert "generator/~/res_block1_11/~/convs2_2" in keys
    by = {s.key: s for s in specs}
    assert by["generator/~/ups_0"].w_shape == (16, 256, 512)
    assert by["generator/~/res_block1_4/~/convs1_2"].w_shape == (7, 128, 128)
    assert by["generator/~/res_block1_4/~/convs1_2"].dilation == 5
    assert by["generator/~/res_block1_4/~/convs2_2"].dilation == 1
    assert by["generator/~/conv1_d_1"].w_shape == (7, 32, 1)


def test_layout_map_is_the_converters():
    """convert_torch_model_to_haiku.py:53-56: rot90(axes=(0,2)) for ups, swapaxes(0,2) for convs."""
    rng = np.random.default_rng(1)
    by = {s.key: s for s in conv_specs(V1)}
    up = by["generator/~/ups_2"]
    w = rng.standard_normal(up.torch_w_shape).astype(np.float32)
    assert np.array_equal(torch_weight_to_haiku(up, w), np.rot90(w, k=1, axes=(0, 2)))
    cv = by["generator/~/res_block1_9/~/convs1_1"]
    w = rng.standard_normal(cv.torch_w_shape).astype(np.float32)
    assert np.array_equal(torch_weight_to_haiku(cv, w), np.swapaxes(w, 0, 2))


def test_roundtrip_and_pickle(tmp_path):
    p = synthetic_params(TINY, 7)
    check_params(TINY, p)
    sd = haiku_to_state_dict(TINY, p)
    q = state_dict_to_haiku(TINY, sd)
    for k in p:
        assert np.array_equal(p[k]["w"], q[k]["w"]) and np.array_equal(p[k]["b"], q[k]["b"])
    # non-contiguous views, as the reference converter pickles them
    views = {k: {"w": np.swapaxes(np.swapaxes(m["w"], 0, 2).copy(), 0, 2), "b": m["b"]} for k, m in p.items()}
    with open(tmp_path / "hk_hifi.pickle", "wb") as f:
        pickle.dump(views, f)
    r = load_haiku_pickle(tmp_path / "hk_hifi.pickle")
    for k in p:
        assert r[k]["w"].flags["C_CONTIGUOUS"] and np.array_equal(r[k]["w"], p[k]["w"])
    save_haiku_pickle(tmp_path / "x.pickle", p)
    assert set(load_haiku_pickle(tmp_path / "x.pickle")) == set(p)


def test_check_params_errors():
    p = synthetic_params(TINY, 7)
    bad = dict(p)
    del bad["generator/~/ups_1"]
    with pytest.raises(ValueError):
        check_params(TINY, bad)
    bad = {k: dict(v) for k, v in p.items()}
    bad["generator/~/ups_1"]["w"] = bad["generator/~/ups_1"]["w"][:-1]
    with pytest.raises(ValueError):
        check_params(TINY, bad)
    with pytest.raises(ValueError):
        HifiganConfig.from_dict({"resblock": "2"})


def test_converter_entry_point_bit_exact_vs_reference_converter(tmp_path, monkeypatch, golden_dir):
    """``python -m vietTTS.hifigan.convert_torch_model_to_haiku --checkpoint-file g_* --config-file config.json``
    (convert_torch_model_to_haiku.py:65-79, scripts/quick_start.sh:7) on a weight-norm checkpoint: the pickle it writes
    under FLAGS.ckpt_dir equals the REFERENCE converter's own output bit for bit (golden minted by oracle/make_golden.py)."""
    import json

    import torch

    from vietTTS.hifigan.convert_torch_model_to_haiku import main as convert_main  # the reference's module path
    from viettts_amd.hifigan.config import FLAGS

    g = np.load(golden_dir / "convert_tiny.npz")
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("SD::")}
    assert any(k.endswith("weight_g") for k in sd)
    monkeypatch.chdir(tmp_path)
    torch.save({"generator": sd}, tmp_path / "g_00000001")
    cfgd = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=32,
                resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=80, sampling_rate=16000)
    (tmp_path / "config.json").write_text(json.dumps(cfgd))
    convert_main(["--checkpoint-file", str(tmp_path / "g_00000001"), "--config-file", str(tmp_path / "config.json")])
    got = load_haiku_pickle(tmp_path / FLAGS.ckpt_dir / "hk_hifi.pickle")
    want = {}
    for k in g.files:
        if k.startswith("HK::"):
            _, key, which = k.split("::")
            want.setdefault(key, {})[which] = g[k]
    assert set(got) == set(want) and len(got) == 78
    for key in want:
        for which in ("w", "b"):
            assert got[key][which].dtype == np.float32
            assert np.array_equal(got[key][which], want[key][which]), (key, which)
    check_params(TINY, got)


def test_resblock2_inventory_and_names():
    """ResBlock2 generators (model.py:54-74): 2 convolutions per block under the names the Haiku model creates
    (res_block1_N: model.py:105; default hk.Conv1D names: model.py:58-66), upstream torch names resblocks.N.convs.Z."""
    specs = conv_specs(TINY2)
    assert len(specs) == 1 + 4 + 12 * 2 + 1
    by = {s.key: s for s in specs}
    a, b = by["generator/~/res_block1_4/~/conv1_d"], by["generator/~/res_block1_4/~/conv1_d_1"]
    assert (a.k, a.dilation, b.dilation, a.torch_prefix, b.torch_prefix) == (5, 2, 6, "resblocks.4.convs.0", "resblocks.4.convs.1")
    params = synthetic_params(TINY2, 1, "scaled")
    check_params(TINY2, params)
    back = state_dict_to_haiku(TINY2, haiku_to_state_dict(TINY2, params))
    assert all(np.array_equal(back[k][n], params[k][n]) for k in params for n in ("w", "b"))
    with pytest.raises(ValueError):
        HifiganConfig(resblock="2").validate()  # ResBlock2 takes two dilations per kernel size
    with pytest.raises(ValueError):
        HifiganConfig(resblock="3").validate()
