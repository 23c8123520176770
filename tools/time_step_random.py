import sys, time, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "pika-zoo_amd"))
from pikazoo_amd import pikazoo_v0
for kw in (dict(is_player2_computer=True), dict(is_player1_computer=True, is_player2_computer=True), dict()):
    env = pikazoo_v0.env(num_envs=65536, device="cuda:0", seed=0, **kw)
    raw = env.unwrapped
    env.reset()
    for _ in range(4):
        raw.step_random(7, k=512)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(16):
        raw.step_random(7, k=512)
    e1.record()
    torch.cuda.synchronize()
    print(kw, "pz_step_random k=512: %.3f us per frame" % (e0.elapsed_time(e1) * 1e3 / (16 * 512)), flush=True)
