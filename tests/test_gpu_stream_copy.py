"""The streaming-rate probe bench.py quotes its roofline fractions against: `ast_stream_copy` copies
bit for bit, leaves its source alone in every mode, covers ragged tails and refuses misaligned arguments."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev(hip):
    from astrild_amd import device
    torch.cuda.set_device(0)
    return device


@pytest.mark.parametrize("count", [4, 1024 * 4 - 4, 1024 * 4 * 9000 + 12])      # floats: below one workgroup, ragged, more than one grid sweep
def test_stream_copy_modes(dev, count):
    from astrild_amd._lib import lib
    src = torch.arange(count, dtype=torch.float32, device="cuda")
    keep = src.clone()
    dst = torch.full((count + 4,), -7.0, dtype=torch.float32, device="cuda")
    assert lib().ast_stream_copy(dev.ptr(dst), dev.ptr(src), count * 4, 0, dev.stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst[:count], keep) and bool((dst[count:] == -7.0).all()) and torch.equal(src, keep)
    dst.fill_(-7.0)
    assert lib().ast_stream_copy(dev.ptr(dst), dev.ptr(src), count * 4, 1, dev.stream()) == 0      # read only
    torch.cuda.synchronize()
    assert bool((dst == -7.0).all()) and torch.equal(src, keep)
    assert lib().ast_stream_copy(dev.ptr(dst), dev.ptr(src), count * 4, 2, dev.stream()) == 0      # write only: the pattern 1, 2, 3, 4
    torch.cuda.synchronize()
    assert torch.equal(dst[:count].view(-1, 4), torch.tensor([1.0, 2.0, 3.0, 4.0], device="cuda").expand(count // 4, 4))
    assert bool((dst[count:] == -7.0).all())


@pytest.mark.parametrize("tune", range(16))
def test_stream_copy_variants_copy_alike(dev, tune):
    from astrild_amd._lib import lib
    count = 1024 * 4 * 3 + 8
    src = torch.arange(count, dtype=torch.float32, device="cuda")
    dst = torch.full((count + 4,), -7.0, dtype=torch.float32, device="cuda")
    assert lib().ast_stream_copy(dev.ptr(dst), dev.ptr(src), count * 4, 256 | (tune << 4), dev.stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst[:count], src) and bool((dst[count:] == -7.0).all())


def test_stream_copy_rejects_bad_arguments(dev):
    from astrild_amd._lib import lib
    a = torch.zeros(64, dtype=torch.float32, device="cuda")
    b = torch.zeros(64, dtype=torch.float32, device="cuda")
    assert lib().ast_stream_copy(dev.ptr(b), dev.ptr(a), 24, 0, dev.stream()) != 0            # not a multiple of 16
    assert lib().ast_stream_copy(b.data_ptr() + 4, dev.ptr(a), 16, 0, dev.stream()) != 0        # misaligned
    assert lib().ast_stream_copy(dev.ptr(b), dev.ptr(a), 16, 3, dev.stream()) != 0            # unknown mode
    assert lib().ast_stream_copy(dev.ptr(b), dev.ptr(a), 16, 16, dev.stream()) != 0           # variant bits without bit 8
    assert lib().ast_stream_copy(dev.ptr(b), dev.ptr(a), 0, 0, dev.stream()) == 0
