import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def poison_allocator(big_mib: int = 768, small_blocks: int = 512):
    """Fill the caching allocator's free pools with NaN patterns (bf16 0x7FC0 = fp32 NaN = a huge int32):
    memory handed out by the following torch.empty() calls then reads as NaN, so a kernel that consumes
    a buffer (or pad columns / workspace rows) nobody wrote fails LOUDLY instead of depending on what a
    previous test or process left in HBM.  Round 2's intermittent one-weight mismatch between two
    world-2 steps only ever showed inside the full pytest process -- the one setting where freshly
    allocated device memory is not zero.  MACAW_NO_POISON=1 switches it off."""
    import torch
    if os.environ.get("MACAW_NO_POISON") or not torch.cuda.is_available():
        return
    big = torch.full((big_mib << 19,), float("nan"), dtype=torch.bfloat16, device="cuda")
    smalls = [torch.full((64 << 10,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(small_blocks)]
    mids = [torch.full((2 << 20,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(32)]
    torch.cuda.synchronize()
    del big, smalls, mids
    poison_side_streams(min(big_mib, 256), min(small_blocks, 128))


def poison_side_streams(big_mib: int = 256, small_blocks: int = 128):
    """the same for the pools of the engine's SIDE streams (engine.DW_SIDE / ENC_SIDE: the caching allocator keeps one pool
    per stream, and what a side-stream launch allocates -- a fresh grad-weight output, that stream's GEMM scratch -- comes
    out of memory the main pool's poison never touched).  The streams are created here if they do not exist yet, so that
    the first backward finds poisoned pools."""
    import torch
    from macaw_llm_amd import engine
    dev = torch.device("cuda", torch.cuda.current_device())
    for reg in (engine.DW_SIDE["streams"], engine.ENC_SIDE["streams"]):
        st = reg.get(dev)
        if st is None:
            st = reg[dev] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            big = torch.full((big_mib << 19,), float("nan"), dtype=torch.bfloat16, device="cuda")
            smalls = [torch.full((64 << 10,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(small_blocks)]
            mids = [torch.full((2 << 20,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(16)]
        st.synchronize()
        del big, smalls, mids


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    if request.node.get_closest_marker("gpu") is not None:
        poison_allocator()
    yield
