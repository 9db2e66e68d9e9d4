"""``sam3.device`` facade (reference: sam3/sam3/device.py)."""
from efficientsam3_amd.device import *  # noqa: F401,F403
from efficientsam3_amd.device import get_device  # noqa: F401
