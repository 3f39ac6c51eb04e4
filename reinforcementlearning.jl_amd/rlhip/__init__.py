"""rlhip -- host-side mirror of the ReinforcementLearning.jl plugin surface for the MI355X-native
rollout + learner hot path.  Everything numeric runs in librlhip.so (HIP, gfx950) through the C ABI
declared in include/rlhip.h; importing this package fails loudly if that library is missing.
"""
from . import _lib  # noqa: F401  (raises ImportError when librlhip.so is absent)
from ._lib import RLHipArgumentError, RLHipError  # noqa: F401
from .envs import (AcrobotRK4Env, CartPoleEnv, ContinuousMountainCarEnv, HipVecEnv, MountainCarEnv,  # noqa: F401
                   PendulumEnv, Space)
from .core import (Agent, BatchStepsPerEpisode, ComposedHook, DeviceEpisodeStats, DoEveryNSteps, EmptyHook,  # noqa: F401
                   PPOAgent, RandomPolicy, StepsPerEpisode, StopAfterNEpisodes, StopAfterNSeconds,
                   StopAfterNSteps, StopIfAll, StopIfAny, TimePerStep, TotalBatchRewardPerEpisode, run,
                   run_fused_dqn, run_fused_ppo)
from .dqn import (DQNLearner, EpsilonGreedyExplorer, GreedyExplorer, HipApproximator,  # noqa: F401
                  QBasedPolicy, TargetNetwork)
from .explorers import (BatchExplorer, GumbelSoftmaxExplorer, UCBExplorer, WeightedExplorer,  # noqa: F401
                        WeightedSoftmaxExplorer)
from .checkpoint import load_checkpoint, load_state_dict, save_checkpoint, state_dict  # noqa: F401
from .timing import disable_debug_timings, enable_debug_timings, timer  # noqa: F401
from .heads import CategoricalNetwork, GaussianNetwork, SoftGaussianNetwork  # noqa: F401
from .ppo import PPOPolicy, PPOTrajectory, make_ppo_cfg  # noqa: F401
from .trajectory import (BatchSampler, CircularArraySARTSTraces, CircularPrioritizedTraces, NStepBatchSampler,  # noqa: F401
                         InsertSampleRatioController, Trajectory)

ABI_VERSION = _lib.lib.rlhip_abi_version()
