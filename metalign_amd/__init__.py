"""metalign_amd — MI355X-native hot path of Metalign (CMash-style containment pre-filter + per-read
taxon assignment / abundance profile) behind the reference's CLI.  See DESIGN.md."""
__version__ = "0.1.0"
