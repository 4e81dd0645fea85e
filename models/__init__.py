"""Import-path aliases so code written against the reference layout (`from models.modelsTF import
WDSRConv3D`, `from models.loss import Losses`, ...) resolves to the MI355X engine in `proba-v_amd/`."""
