"""Host-side mirror of the reference's ``mast3r_slam`` package for the hot path only
(matching, MASt3R inference wrappers, frame data layout, tracker GN, factor-graph GN, dual TSDF).
Everything numeric runs in libmslam_hip.so; see ../mslam_hip.py for the binding layer."""
