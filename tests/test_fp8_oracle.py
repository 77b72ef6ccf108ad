"""CPU pin of the e4m3fn quantiser restated in oracle/ops_ref.py against torch's own float8_e4m3fn cast (the format is not in
the reference: BASELINE.json configs[4] asks for it as an MI355X extension)."""
import torch

from oracle import ops_ref as R


def test_quantize_matches_torch_cast_on_every_code_and_between_them():
    codes = torch.arange(256, dtype=torch.uint8)
    vals = codes.view(torch.float8_e4m3fn).float()
    finite = vals[torch.isfinite(vals)]
    assert finite.numel() == 254 and finite.abs().max() == 448.0
    assert torch.equal(R.quantize_e4m3(finite), finite)  # grid points are fixed points
    g = torch.Generator().manual_seed(0)
    x = torch.cat([(torch.rand(20000, generator=g) * 2 - 1) * 440, (torch.rand(20000, generator=g) * 2 - 1) * 0.05, torch.randn(20000, generator=g)])
    assert torch.equal(R.quantize_e4m3(x), x.to(torch.float8_e4m3fn).float())
    s = finite.sort().values  # exact midpoints between neighbouring codes: ties go to the even mantissa
    mid = (s[1:] + s[:-1]) / 2
    assert torch.equal(R.quantize_e4m3(mid), mid.to(torch.float8_e4m3fn).float())


def test_quantize_saturates_and_scales():
    x = torch.tensor([1e4, -1e4, 449.0, 3.0, -0.3])
    assert R.quantize_e4m3(x).tolist()[:3] == [448.0, -448.0, 448.0]
    assert torch.equal(R.quantize_e4m3(x[3:], 2.0), (x[3:] * 2).to(torch.float8_e4m3fn).float())
    assert torch.equal(R.e4m3_bits(torch.tensor([448.0, -448.0, 0.0, 2.0 ** -9])), torch.tensor([0x7E, 0xFE, 0x00, 0x01], dtype=torch.uint8))
