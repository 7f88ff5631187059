"""istnet_amd -- MI355X-native (gfx950) point-cloud hot path of IST-Net.

The directory is named ``ist-net_amd``; import it through the ``istnet_amd`` alias module at
the repository root.
"""
__version__ = "0.1.0"
