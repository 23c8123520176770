import sys
sys.path.insert(0, 'pika-zoo_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from oracle import pz_oracle as po
from test_gpu_parity import make_env, cpu
n = 4096
kw = dict(is_player1_computer=True, is_player2_computer=True, winning_score=3)
for fmt in ("int32", "packed"):
    env = make_env(num_envs=n, seed=3, env_id_base=50, state_format=fmt, **kw)
    raw = env.unwrapped
    ref = po.OracleEnv(n, po.make_config(seed=3, env_id_base=50, **kw), nthreads=8)
    env.reset(), ref.reset()
    env.step_random(5, t0=0, k=40), ref.rollout_random(5, 0, 40)
    g = torch.Generator(device="cpu").manual_seed(1)
    st = raw.state.cpu()
    st[29] = torch.randint(-200, 201, (n,), generator=g, dtype=torch.int32)
    st[28] = torch.randint(-20, 21, (n,), generator=g, dtype=torch.int32)
    st[3, ::2] = 1
    st[1, ::2] = 200
    st[26, ::2] = st[0, ::2] + 10
    st[27, ::2] = 190
    raw.set_state(st.to(raw.device))
    back = cpu(raw.state)
    print(fmt, "set_state round trip equal:", np.array_equal(back, st.numpy()))
    ref.state[:] = st.numpy()
    acts = raw.random_actions(5, 40)
    env.step(acts)
    ref.step(cpu(acts["player_1"]), cpu(acts["player_2"]))
    hs = cpu(raw.state)
    diff = np.argwhere(hs != ref.state)
    print(fmt, "mismatches after one step:", len(diff))
    for f, l in diff[:12]:
        print("   field", po.FIELD_NAMES[f], "lane", l, "hip", hs[f, l], "oracle", ref.state[f, l], "| planted ball", st[26:30, l].tolist(), "p1", st[0:4, l].tolist(), "p2", st[13:17, l].tolist())
