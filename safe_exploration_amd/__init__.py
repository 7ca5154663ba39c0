"""MI355X-native CEM safe-MPC hot path.  Importing never touches the GPU; calling does."""
__version__ = '0.1'
