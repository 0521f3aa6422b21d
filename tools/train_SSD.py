"""Active-learning driver for SSD300-VGG16 + MEH/HUA (reference tools/train_SSD.py): the same cycle loop as
tools/train_RetinaNet.py with the SSD config, detector, runner (MyEpochBasedRunnerLSSD) and 300x300 inputs."""
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))
from train_RetinaNet import main  # noqa: E402

if __name__ == '__main__':
    main('configs/_base_/Config_SSD.py', 300)
