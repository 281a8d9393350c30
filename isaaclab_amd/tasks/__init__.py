"""Task definitions on the hot path (mirror of the relevant parts of ``isaaclab_tasks``)."""
