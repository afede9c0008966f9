"""RegNet-Y 800MF oracle vs the independent implementation shipped with `transformers`
(built from a local config, no download): same weights mapped name to name must give the
same pooled features.  This is the structural cross-check of the timm-side arithmetic."""
import numpy as np
import pytest
import torch

from oracle import stes_oracle as O


def _map_key(k):
    # oracle/timm-style name -> transformers name
    if k.startswith("stem."):
        return k.replace("stem.conv", "embedder.embedder.convolution").replace("stem.bn", "embedder.embedder.normalization")
    s, b, rest = k.split(".", 2)
    pre = f"encoder.stages.{int(s[1]) - 1}.layers.{int(b[1:]) - 1}."
    for o, h in (("conv1", "layer.0"), ("conv2", "layer.1"), ("conv3", "layer.3"), ("downsample", "shortcut")):
        if rest.startswith(o + "."):
            return pre + rest.replace(o + ".conv", h + ".convolution").replace(o + ".bn", h + ".normalization")
    if rest.startswith("se."):
        return pre + rest.replace("se.fc1", "layer.2.attention.0").replace("se.fc2", "layer.2.attention.2")
    raise KeyError(k)


def test_regnety008_matches_transformers_twin():
    tr = pytest.importorskip("transformers")
    cfg = tr.RegNetConfig(embedding_size=32, hidden_sizes=[64, 128, 320, 768], depths=[1, 3, 8, 2],
                          groups_width=16, layer_type="y")
    hf = tr.RegNetModel(cfg).eval()
    net = O.RegNetY("regnety_008")
    net.head.fc = torch.nn.Identity()
    O.randomize_(net, seed=9)
    net.eval()
    sd = {}
    for k, v in net.state_dict().items():
        sd[_map_key(k)] = v
    missing = set(hf.state_dict().keys()) - set(sd.keys())
    assert not missing, sorted(missing)[:5]
    hf.load_state_dict(sd, strict=True)
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == 5_494_168
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a = net(x)
        b = hf(x).pooler_output.flatten(1)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=1e-5)
