"""DeterministicBandit lanes of the oracle (src/envs/bandits.rs:66-77, 109-116): reward = the chosen arm's value, every
step terminates and starts a new episode, the observation is the one-hot of the single state."""
import numpy as np

import oracle as O


def test_bandit_lanes_follow_bandit_step():
    sim = O.BanditLaneSim(6, values=(0.25, 1.5))
    assert sim.D == 5
    obs0 = sim.observe()
    assert np.array_equal(obs0, np.repeat(np.array([[1.0], [0], [0], [0], [0]], dtype=np.float32), 6, axis=1))
    _, _, rc0 = sim.get_state()
    actions = np.array([0, 1, 1, 0, 1, 0], dtype=np.uint8)
    for k in range(3):
        reward, flag, obs, _ = sim.step(actions)
        assert np.array_equal(reward, np.where(actions == 1, np.float32(1.5), np.float32(0.25)))
        assert (flag == O.TERMINATE).all()            # Successor::Terminate (bandits.rs:76)
        assert np.array_equal(obs, obs0)              # the new episode's observation
        state, _, rc = sim.get_state()
        assert (state == 0).all() and np.array_equal(rc, rc0 + k + 1)
