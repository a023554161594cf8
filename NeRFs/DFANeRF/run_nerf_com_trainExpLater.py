"""Drop-in for the reference's driver: `python NeRFs/DFANeRF/run_nerf_com_trainExpLater.py --config ... <flags>`
exactly as scripts/train_obama.sh and scripts/test_obama.sh invoke it."""
import _bootstrap  # noqa: F401
from dfanerf.run_nerf import *  # noqa: F401,F403
from dfanerf.run_nerf import (FrameRenderer, calc_volume_weights, composite_function, config_parser,  # noqa: F401
                              create_nerf, encode_signal, encode_signal_torso, euler2rot, parse_config_file,
                              pose_to_euler_trans, render_rays, rot_to_euler, run_network, train)

if __name__ == '__main__':
    train()
