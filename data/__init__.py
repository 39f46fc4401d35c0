"""Import-path shim: `from data.audio import Audio` resolves to the MI355X-native mel path."""
