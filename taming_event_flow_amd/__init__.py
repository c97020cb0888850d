"""MI355X-native implementation of the tudelft/taming_event_flow training hot path.

Sub-modules mirror the reference layout (``loss/flow.py`` ...) so that, with this directory on
``sys.path``, ``from loss.flow import *`` resolves to the HIP-backed modules.
"""

__version__ = "0.1.0"
