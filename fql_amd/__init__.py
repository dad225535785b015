"""fql_amd: MI355X-native (gfx950) Flow Q-Learning gradient step behind the reference's FQLAgent API.

`agents` mirrors the reference registry (agents/__init__.py:10-19) for the one agent on the built path.
"""
from .agent import FQLAgent, INFO_KEYS, NOISE_KEYS
from .config import get_config
from .datasets import Dataset, ReplayBuffer

agents = dict(fql=FQLAgent)

__all__ = ['FQLAgent', 'get_config', 'agents', 'INFO_KEYS', 'NOISE_KEYS', 'Dataset', 'ReplayBuffer']
