"""Import-compatible alias so that the reference's documented entry point
(`from perspective2d import PerspectiveFields`, reference perspective2d/__init__.py:1,
README.md:98-112) resolves to the MI355X implementation."""
from perspectivefields_amd.perspectivefields import PerspectiveFields, model_zoo  # noqa: F401
