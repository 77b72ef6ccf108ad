"""ORACLE (test infrastructure — never imported by the product path).

Plain-torch CPU restatement of the BResNet-50 variant of BASELINE.json configs[3]:
`pytorch_tools.models.resnet50(stem_type="deep", antialias=True, attn_type="eca", norm_layer="inplaceabn", norm_act="leaky_relu",
drop_rate=0.2, drop_connect_rate=0.2)` (configs/_old_configs/_first_attempts/BResNet50_encoder.yaml:41-51) with weight
standardisation of every conv (train.py:66-67, `weight_standardization: True` yaml:59).  pytorch_tools is not vendored and
cannot be imported here, so the block definitions are restated from its published source as recalled in SURVEY.md Appendix C —
UPSTREAM-RECALLED, parity unpinned (the reference holds no vectors for it either).  What is restated:
  deep stem      conv3x3(3, 32, s2) ABN, conv3x3(32, 32) ABN, conv3x3(32, 64), bn1 = ABN(64); anti-aliased pool = maxpool 3x3/1 + BlurPool
  bottleneck     conv1x1 ABN, conv3x3 (stride 1 when anti-aliased) ABN [BlurPool if the block strides], conv1x1 ABN(identity), ECA(k=3),
                 drop-connect on the residual branch (rate = drop_connect_rate * block index / number of blocks), + shortcut, leaky ReLU;
                 shortcut of a striding / widening block: [AvgPool 2x2 if it strides] conv1x1 ABN(identity)
  head           GAP, dropout(drop_rate), FC
  ABN            BatchNorm (eps 1e-5) + leaky ReLU 0.01;  WS: (w - mean) / sqrt(var + 1e-5) per output channel
Randomness (drop-connect / dropout keep masks) is an INPUT: `masks` = {"dc": [per block [N] scale or None], "do": [N, 2048] scale or None}.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

LEAKY = 0.01


def ws(w, eps=1e-5):
    var, mean = torch.var_mean(w, dim=(1, 2, 3), keepdim=True, unbiased=False)
    return (w - mean) / torch.sqrt(var + eps)


class WSConv(nn.Conv2d):
    def __init__(self, cin, cout, k, stride=1, standardize=True):
        super().__init__(cin, cout, k, stride, k // 2, bias=False)
        self.standardize = standardize

    def forward(self, x):
        return F.conv2d(x, ws(self.weight) if self.standardize else self.weight, None, self.stride, self.padding)


class ABN(nn.BatchNorm2d):
    def __init__(self, c, act="leaky_relu"):
        super().__init__(c, eps=1e-5, momentum=0.1)
        self.act = act

    def forward(self, x):
        y = super().forward(x)
        return F.leaky_relu(y, LEAKY) if self.act == "leaky_relu" else y


class BlurPool(nn.Module):
    def forward(self, x):
        C = x.shape[1]
        f = torch.tensor([1.0, 2.0, 1.0], dtype=x.dtype)
        k = (f[:, None] * f[None, :] / 16.0)[None, None].repeat(C, 1, 1, 1)
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), k, stride=2, groups=C)


class ECA(nn.Module):
    def __init__(self, k=3):
        super().__init__()
        self.conv = nn.Conv1d(1, 1, k, padding=k // 2, bias=False)

    def forward(self, x):
        N, C = x.shape[:2]
        y = self.conv(x.mean((2, 3)).view(N, 1, C)).view(N, C, 1, 1)
        return x * torch.sigmoid(y)


class Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, downsample, standardize):
        super().__init__()
        self.conv1, self.bn1 = WSConv(cin, planes, 1, 1, standardize), ABN(planes)
        self.conv2, self.bn2 = WSConv(planes, planes, 3, 1, standardize), ABN(planes)  # anti-aliased: the stride moves to the blur
        self.blurpool = BlurPool() if stride == 2 else None
        self.conv3, self.bn3 = WSConv(planes, planes * 4, 1, 1, standardize), ABN(planes * 4, act="identity")
        self.se_module = ECA(3)
        self.downsample = downsample

    def forward(self, x, keep):
        out = self.bn1(self.conv1(x))
        out = self.bn2(self.conv2(out))
        if self.blurpool is not None:
            out = self.blurpool(out)
        out = self.se_module(self.bn3(self.conv3(out)))
        if keep is not None:
            out = out * keep.view(-1, 1, 1, 1)
        sc = x if self.downsample is None else self.downsample(x)
        return F.leaky_relu(out + sc, LEAKY)


class BResNet50Ref(nn.Module):
    def __init__(self, num_classes=1000, standardize=True):
        super().__init__()
        S = standardize
        self.conv1 = nn.Sequential(WSConv(3, 32, 3, 2, S), ABN(32), WSConv(32, 32, 3, 1, S), ABN(32), WSConv(32, 64, 3, 1, S))
        self.bn1 = ABN(64)
        self.maxpool = nn.Sequential(nn.MaxPool2d(3, 1, 1), BlurPool())
        cin = 64
        for li, (nb, planes) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), 1):
            blocks = []
            for i in range(nb):
                stride = 2 if (i == 0 and li > 1) else 1
                ds = None
                if i == 0:
                    mods = ([("blur", nn.AvgPool2d(2, 2))] if stride == 2 else []) + [("0", WSConv(cin, planes * 4, 1, 1, S)), ("1", ABN(planes * 4, act="identity"))]
                    ds = nn.Sequential()
                    for n, m in mods:
                        ds.add_module(n, m)
                blocks.append(Bottleneck(cin, planes, stride, ds, S))
                cin = planes * 4
            setattr(self, f"layer{li}", nn.ModuleList(blocks))
        self.fc = nn.Linear(2048, num_classes)

    def blocks(self):
        return [b for li in range(1, 5) for b in getattr(self, f"layer{li}")]

    def forward(self, x, masks=None):
        x = self.maxpool(self.bn1(self.conv1(x)))
        for i, b in enumerate(self.blocks()):
            x = b(x, None if masks is None else masks["dc"][i])
        x = x.mean((2, 3))
        if masks is not None and masks.get("do") is not None:
            x = x * masks["do"]
        return self.fc(x)
