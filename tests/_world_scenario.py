"""Scripted imagined-rollout scenario, run against the reference's WorldVecEnv (oracle/gen_golden.py)
and this repo's (tests/test_world_env.py).  All module objects are passed in."""
import numpy as np
import torch


def run(M, device="cpu", world_kwargs=None):
    """M: namespace with Env, T (transforms), Replay, ds, Sample, factory_cls, TrainingModule, Ensemble,
    WorldVecEnv.  world_kwargs: extra keywords for this repo's WorldVecEnv -- a callable value is called with the
    scenario's env (e.g. ``{"batched_reward_func": lambda env: env.batched_reward_func}``)."""
    rec = {}
    env = M.Env()
    T = M.T
    N, tstep, tau = env.N, env.cfg_steps * env.dt, 3

    # transforms as in pdecontrol/mbrl/mbrl.py:146-187 (observation scaling frozen with fixed bounds)
    oscaling = T.ScaleTransform(bounds=(np.full((1, 1, 1), -3.0, np.float32), np.full((1, 1, 1), 3.0, np.float32)),
                                batched=True, aggregate=True, frozen=True)
    forcing = T.BatchTransform(env.forcing)
    low = np.asarray(env.action_space.low)[np.newaxis, ...]
    high = np.asarray(env.action_space.high)[np.newaxis, ...]
    lo = np.squeeze(forcing(low), axis=0)
    hi = np.squeeze(forcing(high), axis=0)
    pdescaling = T.BatchTransform(T.ScaleTransform(bounds=(lo, hi), scale=(-1, 1), aggregate=True, frozen=True))
    world_sensor = T.BatchTransform(T.SensorTransform(stride=1))
    replay_to_world = T.SampleTransform([oscaling, world_sensor], [forcing, pdescaling, world_sensor])

    # replay with two finished episodes of a smooth synthetic field
    rp = M.Replay()
    rs = np.random.RandomState(5)
    x = np.linspace(0, 2 * np.pi, N, endpoint=False)
    for ep_len in (7, 9):
        phase = rs.uniform(0, 6)
        for t in range(ep_len):
            mk = lambda tt: (np.sin(x + phase + 0.3 * tt) + 0.5 * np.cos(2 * x - 0.2 * tt)).astype(np.float32)[None, :]
            act = rs.uniform(-1, 1, (1, 4)).astype(np.float32)
            rp.add([M.Sample(mk(t), act, mk(t + 1), np.float32(-1.0), False, t == ep_len - 1, np.int32(t + 1))])

    # ensemble of two freshly seeded surrogates
    modules = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        f = M.factory_cls()
        sur = f.surrogate(delta=tstep, dscaling=None, tau=tau, **f.model())
        modules.append(M.TrainingModule(surrogate=sur, loss=torch.nn.MSELoss(reduction="none"), tstep=tstep, delta=tstep,
                                        tau=tau, tbtt=10).to(device))
    ensemble = M.Ensemble(modules, num_elites=2)

    world = M.WorldVecEnv(surrogate=ensemble, observation_space=env.observation_space, action_space=env.action_space,
                          max_episode_steps=env.max_episode_steps, stransf=replay_to_world.Inverse,
                          reward_func=env.reward_func, num_envs=4, horizon=3, tstep=tstep,
                          **{k: (v(env) if callable(v) else v) for k, v in (world_kwargs or {}).items()})
    rec_world = world
    rec["single_action_shape"] = np.asarray(world.single_action_space.shape)
    rec["single_obs_shape"] = np.asarray(world.single_observation_space.shape)
    rec["action_low"] = np.asarray(world.single_action_space.low)

    starting = M.ds.StartingStateDataset(data=rp.data, length=tau, stride=1, bootstrapping=False, stransf=replay_to_world)
    rec["starting_len"] = np.asarray(len(starting))
    world.setup(starting)
    torch.manual_seed(123)
    np.random.seed(321)
    obs, info = world.reset(return_info=True)
    rec["reset_obs"], rec["reset_step"] = np.asarray(obs), np.asarray(info["step"])
    ars = np.random.RandomState(9)
    for k in range(5):
        a = ars.uniform(-1, 1, (4,) + tuple(world.single_action_space.shape)).astype(np.float32)
        world.step_async(a)
        obs, rew, term, trunc, infos = world.step_wait()
        rec[f"s{k}_obs"], rec[f"s{k}_rew"] = np.asarray(obs), np.asarray(rew)
        rec[f"s{k}_trunc"], rec[f"s{k}_step"] = np.asarray(trunc), np.asarray(infos["step"])
        rec[f"s{k}_has_final"] = np.asarray("final_observation" in infos)
        if "final_observation" in infos:
            rec[f"s{k}_final"] = np.asarray(infos["final_observation"])
    run.last_world = rec_world
    return rec
