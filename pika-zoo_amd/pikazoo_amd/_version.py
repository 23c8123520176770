"""Version of the MI355X build (independent of the reference's release tags)."""
VERSION = (0, 1, 0)
__version__ = ".".join(str(part) for part in VERSION)
