"""MI355X-native bundle adjustment for visual_marker_mapping's TagReconstructor hot path."""
